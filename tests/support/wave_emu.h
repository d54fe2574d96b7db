// wave_emu.h -- CPU emulation of one 64-lane wavefront for debugging the
// wave-level code of rimphony_amd/csrc (tests only; never part of the product).
// Each lane is a host thread; every cross-lane primitive is a collective built
// from two barriers, which is valid because the kernels only use them in
// wave-uniform control flow.
#ifndef RIM_WAVE_EMU_H
#define RIM_WAVE_EMU_H
#include <pthread.h>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)

namespace rim {

struct EmuWave {
    pthread_barrier_t bar;
    unsigned long long slot[64];
};
inline EmuWave &emu_wave() { static EmuWave w; return w; }
inline int &emu_lane_ref() { static thread_local int lane = 0; return lane; }

inline void wv_sync() { pthread_barrier_wait(&emu_wave().bar); }
inline int wv_lane() { return emu_lane_ref(); }

inline unsigned long long emu_exchange(unsigned long long v, int src)
{
    EmuWave &w = emu_wave();
    w.slot[wv_lane()] = v;
    pthread_barrier_wait(&w.bar);
    const unsigned long long r = w.slot[src & 63];
    pthread_barrier_wait(&w.bar);
    return r;
}
inline int wv_readlane(int v, int srclane) { return (int) (unsigned) emu_exchange((unsigned) v, srclane); }
inline int wv_readfirstlane(int v) { return wv_readlane(v, 0); }
inline double wv_shfl_xor(double v, int m)
{
    unsigned long long u;
    std::memcpy(&u, &v, 8);
    u = emu_exchange(u, wv_lane() ^ m);
    std::memcpy(&v, &u, 8);
    return v;
}
inline int wv_shfl_xor(int v, int m) { return (int) (unsigned) emu_exchange((unsigned) v, wv_lane() ^ m); }
inline unsigned long long wv_ballot(bool p)
{
    EmuWave &w = emu_wave();
    w.slot[wv_lane()] = p ? 1ull : 0ull;
    pthread_barrier_wait(&w.bar);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (w.slot[i] & 1ull) << i;
    pthread_barrier_wait(&w.bar);
    return m;
}

}  // namespace rim
#endif
