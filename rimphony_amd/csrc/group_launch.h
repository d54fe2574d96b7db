// group_launch.h -- what rimphony_hip.hip (host side) and rimphony_group.hip (the kernel) agree on.
#ifndef RIM_GROUP_LAUNCH_H
#define RIM_GROUP_LAUNCH_H

#include "symphony_group.h"
#include "heyvaerts_group.h"
#include "coop_common.h"

// waves per SIMD the register allocator must leave room for.  Symphony groups: 5 (96 VGPRs, 20 waves x 7.6 KB of LDS per
// CU).  Measured on one MI355X, 32768 six-coefficient power-law rows, after the uniform hints of wave_qag_group:
// 3 -> 1416 ms, 4 -> 1226 ms, 5 -> 1156 ms (thermal, both pitchy tables and the j_I/alpha_I pair: 5 is 4-5 % faster
// than 4 as well); 6 does not fit the LDS without shortening the lists (tried with 16 entries: 2.8 x slower).
// Before the hints 4 was the fastest (5: +6 %).
#ifndef RIM_GROUP_WAVES
#define RIM_GROUP_WAVES 5
#endif
#ifndef RIM_HEY_GROUP_WAVES
#define RIM_HEY_GROUP_WAVES 4
#endif
// idle waves that stay per wave that still owns a task at the end of a launch (the surplus leaves: more helpers than a
// round has entries only add polling traffic)
#ifndef RIM_GROUP_HELPERS_PER_OWNER
#define RIM_GROUP_HELPERS_PER_OWNER 64u
#endif
// entries of one published round: up to 62 lanes x RIM_GROUP classes
#define RIM_GROUP_ENTRIES 248

// Board slot of the group kernel (cooperative tail; protocol: coop_common.h).  An entry = one merged request
// (n, lobe, member mask); its result = per member a value, status bits and the integrand samples the member's
// quadrature consumed.
struct GroupSlot {
    unsigned long long claim;       // (seq << 32) | (count << 8) | next
    unsigned long long point;
    unsigned done;
    unsigned slots;                 // the group's members (symphony_group.h: group_slot)
    unsigned long long req_n[256];  // bit patterns of doubles: every access is an agent-scope atomic
    int req_tag[256];               // lobe | member mask << 1
    unsigned long long res[256 * RIM_GROUP];
    int res_status[256];            // 8 status bits per member
    unsigned res_samples[256 * RIM_GROUP];
};

struct GroupArgs {
    SymArgs base;                   // slot[] / nslots / board / spill of the base are not used by the group kernel
    int ngroups;
    unsigned gslots[2];             // members of group 0 and 1, 4 bits per member (output slots 0..5)
    int gnmem[2];
    GroupSlot *gboard;              // [gridDim.x]
    double *gspill;                 // [gridDim.x][P::SPILL_PER_WAVE]
    int coop;                       // cooperative tail on
    unsigned long long *prof;       // -DRIM_PROF builds: [gridDim.x][32] region timers / hit counters (else null)
};

// faraday = 0: the Symphony groups ({j_I, alpha_I, j_Q, alpha_Q}, {j_V, alpha_V}); 1: the Faraday pair {rho_Q, rho_V}
const void *rim_group_kernel(int kind, int faraday);
int rim_group_waves(int faraday);
int rim_group_launch(int kind, int faraday, unsigned grid, hipStream_t st, const GroupArgs &ga);

#endif
