"""Builds scripts/exp/libstamp.so: libmudiff_hip.so with cycle stamps (s_memtime) at the phase boundaries of the 3x3 conv
kernel, for scripts/stamp_conv.py.  The instrumented source is generated from csrc/conv_mfma.hip; nothing is shipped.
    python scripts/build_stamp_lib.py && MUDIFF_HIP_LIB=scripts/exp/libstamp.so python scripts/stamp_conv.py 16"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mu-diff_amd', 'csrc')
s = open(os.path.join(CSRC, 'conv_mfma.hip')).read()
s = s.replace("template <int KS, int MT, int WM, int WN, bool DUAL = false>\nstruct CmGeo {", '''__device__ unsigned long long g_stamps[64 * 64];
extern "C" int mud_debug_read_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#define STAMP(i) do { if (tid == 0 && blockIdx.x < 64 && (i) < 64) g_stamps[blockIdx.x * 64 + (i)] = ((i) == 58 || (i) == 59) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); } while (0)
template <int KS, int MT, int WM, int WN, bool DUAL = false>
struct CmGeo {''', 1)
assert "define STAMP" in s
k0, k1 = s.index("void k_conv_mfma(mud_conv_args a"), s.index("// Variant for ks == 1")
b = s[k0:k1]


def sub(old, new):
    global b
    assert old in b, old
    b = b.replace(old, new, 1)


sub("  const int wm = wave % WM, wn = wave / WM;     // wave grid: WM along pixel rows, WN along 64-channel tiles",
    "  const int wm = wave % WM, wn = wave / WM;\n  STAMP(0);")
sub("  store_a(kc0, smem + (kc0 & 1) * G::BUF);\n  __syncthreads();", "  store_a(kc0, smem + (kc0 & 1) * G::BUF);\n  __syncthreads();\n  STAMP(1); STAMP(58);")
sub("    const bool more = kc + 1 < nchunks;\n#pragma unroll\n    for (int g = 0; g < G::NG; ++g) {", "    const bool more = kc + 1 < nchunks;\n    STAMP(2 + kc);\n#pragma unroll\n    for (int g = 0; g < G::NG; ++g) {")
sub("      __syncthreads();                          // DMA of group gg+1 landed (vmcnt drained by the fence) and is visible to all waves",
    "      if (kc == 2) STAMP(40 + 2 * g);\n      asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\" ::: \"memory\");\n      if (kc == 2) STAMP(46 + g);\n      __syncthreads();\n      if (kc == 2) STAMP(41 + 2 * g);")
sub("  // ---- epilogue: D[row = pixel (reg&3)+8*(reg>>2)+4*hh][col = channel r]", "  STAMP(60); STAMP(59);\n  // ---- epilogue")
sub("  if (a.stats) {\n    __syncthreads();\n    if (tid < 128 * WN) {", "  STAMP(61);\n  if (a.stats) {\n    __syncthreads();\n    if (tid < 128 * WN) {")
b = b[:b.rstrip().rfind("}")] + "  STAMP(62);\n}\n\n"
src = os.path.join(CSRC, '_conv_stamp_tmp.hip')
open(src, 'w').write(s[:k0] + b + s[k1:])
out = os.path.join(ROOT, 'scripts', 'exp')
os.makedirs(out, exist_ok=True)
try:
    subprocess.check_call(['make', '-C', CSRC])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-function', '-c', src,
                           '-o', '/tmp/_conv_stamp.o'])
    objs = [os.path.join(CSRC, o) for o in ('elementwise.o', 'groupnorm.o', 'dense.o', 'conv_direct.o', 'fir.o', 'attention.o')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-fPIC', '-std=c++17', '-DMUD_BUILD_FLAGS="stamps"', '-c', os.path.join(CSRC, 'api.cpp'), '-o', '/tmp/_api_stamp.o'])
    objs.append('/tmp/_api_stamp.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '/tmp/_conv_stamp.o', '-o',
                           os.path.join(out, 'libstamp.so')])
finally:
    os.remove(src)
print('built', os.path.join(out, 'libstamp.so'))
