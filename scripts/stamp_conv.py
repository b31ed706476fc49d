"""Phase breakdown of the 3x3 conv kernel from in-kernel cycle stamps.  Needs the instrumented build:
    python scripts/build_stamp_lib.py && MUDIFF_HIP_LIB=scripts/exp/libstamp.so python scripts/stamp_conv.py [batch]"""
import ctypes, sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import numpy as np, torch
import mudiff_hip
from mudiff_hip import ops
lib = mudiff_hip.load()
lib.mud_debug_read_stamps.restype = ctypes.c_int
lib.mud_debug_read_stamps.argtypes = [ctypes.c_void_p]
B, dev = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 'cuda:0'
PLAN = sys.argv[2] if len(sys.argv) > 2 else '16x3'      # 16x3 | fp8x
for H, Cin, Cout, res in [(256, 64, 64, True), (256, 64, 64, False), (128, 128, 128, True), (64, 256, 256, False), (256, 320, 64, False), (256, 128, 64, False)]:
    x = ops.View(torch.randn(B, H, H, Cin, device=dev), B, H, H, Cin)
    wt = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9)
    kwp = {}
    if PLAN == 'fp8x' and ops.conv_prec_supported(ops.View(torch.empty(B, H, H, Cin, device=dev), B, H, H, Cin), Cout, ops.PRO_AFFINE_SILU, ops.PREC_FP8X):
        kwp = dict(prec=ops.PREC_FP8X, w_exp=ops.fp8x_weight_exponent(wt))
    w = ops.pack_conv_weight(wt, **kwp)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    r = ops.View(torch.randn(B, H, H, Cout, device=dev), B, H, H, Cout) if res else None
    arena = ops.StatsArena(dev)
    out = ops.View.empty(B, H, H, Cout, dev, arena)
    for _ in range(3):
        ops.conv(x, w, 3, Cout, mfma=True, pro=(sc, sh, ops.PRO_AFFINE_SILU), res=r, out=out, **kwp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv(x, w, 3, Cout, mfma=True, pro=(sc, sh, ops.PRO_AFFINE_SILU), res=r, out=out, **kwp)
    e1.record(); torch.cuda.synchronize()
    print(f'   back-to-back launches: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per conv (incl. split-K epilogue if any)')
    buf = np.zeros(64 * 64, dtype=np.uint64)
    assert lib.mud_debug_read_stamps(buf.ctypes.data) == 0
    st = buf.reshape(64, 64).astype(np.int64)
    nch = (Cin + 15) // 16
    tot = st[:, 62] - st[:, 0]
    pro = st[:, 1] - st[:, 0]
    loop = st[:, 60] - st[:, 1]
    epi = st[:, 61] - st[:, 60]
    tail = st[:, 62] - st[:, 61]
    chunk = (st[:, 2 + nch - 1] - st[:, 2]) / max(nch - 1, 1)
    clk = (st[:, 60] - st[:, 1]) / np.maximum(st[:, 59] - st[:, 58], 1) * 100.0      # MHz: s_memtime ticks per 100 MHz realtime tick
    print(f'   in-kernel clock over the K loop: median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})')
    f = lambda a: f'{np.median(a):8.0f}'
    grp = [int(np.median(st[:, 40 + i] - (st[:, 2 + 2] if i == 0 else st[:, 39 + i]))) for i in range(6)]
    print('   chunk 2: [group MFMAs+staging, barrier wait] x3 =', grp)
    drain = [int(np.median(st[:, 46 + g] - st[:, 40 + 2 * g])) for g in range(3)]
    skew = [int(np.median(st[:, 41 + 2 * g] - st[:, 46 + g])) for g in range(3)]
    print('   of the barrier wait (wave 0): own memory drain (s_waitcnt vmcnt(0) lgkmcnt(0)) =', drain, ' waiting for the other waves =', skew)
    print(f'{H}^2 {Cin}->{Cout} res={int(res)} plan {"fp8x" if kwp else "16x3"}: cycles(median over 64 blocks, 100 MHz counter?) total{f(tot)} prologue{f(pro)} loop{f(loop)} (per chunk{f(chunk)}, {nch} chunks) epilogue{f(epi)} stats+end{f(tail)}')
