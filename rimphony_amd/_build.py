"""Build helpers: compile the HIP library (gfx950) and, for tests, the CPU oracle.

The product library is librimphony_hip.so, built in-tree with hipcc.  Flags that
matter for correctness (DESIGN.md "Rounding contract"):
  -ffp-contract=off          only the explicit fma() calls fuse, exactly as in the oracle
  -mllvm -disable-machine-licm
      the integrand contains ~300 distinct fp64 literals; gfx9 VOP3 cannot encode
      64-bit literals, and MachineLICM otherwise hoists their materialisation out
      of the quadrature loops into ~200 long-lived VGPRs (256 VGPR + AGPR spills,
      1 wave/SIMD).  With hoisting off the constants are rebuilt by the scalar unit
      next to their use and the Symphony kernel fits 80 VGPRs (6 waves/SIMD).
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rimphony_amd", "csrc")
LIB = os.path.join(ROOT, "rimphony_amd", "librimphony_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")

HIPCC_FLAGS = [
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
    # host side of the translation units (the pkgw_bessel_j / pkgw_bessel_dj seam is the HOST build of
    # dev_bessel.h): inline fma instead of calls into libm, as the oracle's Makefile does
    "-Xarch_host", "-mfma",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hip_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "rimphony_hip.h"))
    return srcs


def source_id():
    """16 hex digits identifying the kernel sources of this tree (csrc/*.h, *.hip, the C-ABI header): profiles record
    it so that bench.py only quotes a PMC figure for the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for path in hip_sources():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_hip(force=False, verbose=False):
    """Compile librimphony_hip.so for gfx950 (cross-compiles without a GPU)."""
    srcs = hip_sources()
    if not force and _newer(LIB, srcs):
        return LIB
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build librimphony_hip.so")
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, "rimphony_hip.hip"), os.path.join(CSRC, "rimphony_diag.hip"),
                                    os.path.join(CSRC, "rimphony_group.hip"), os.path.join(CSRC, "rimphony_multi.hip"),
                                    "-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=ROOT)
    return LIB


def build_oracle(force=False):
    """Compile the CPU oracle (test infrastructure; never loaded by the product)."""
    if force:
        subprocess.run(["make", "-C", ORACLE_DIR, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE_DIR, "liboracle.so")


def build_test_support():
    """Host build of the scalar device functions (CPU tests only)."""
    src = os.path.join(ROOT, "tests", "support", "devfn_host.cpp")
    out = os.path.join(ROOT, "tests", "support", "devfn_host.so")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if _newer(out, deps):
        return out
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-math-errno",
                    "-mfma", "-msse4.1", "-shared", src, "-o", out], check=True)
    return out
