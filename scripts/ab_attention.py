"""In-process A/B of library builds on the attention kernel (N = 4096 keys, C = 256, the config-2 shape), interleaved rounds.
    python scripts/ab_attention.py B rounds lib_a.so lib_b.so ...        (first library = baseline; MUDIFF_ALLOW_VARIANT not needed: raw ctypes)"""
import ctypes as C, os, sys
sys.path[:0] = ['/root/repo', '/root/repo/mu-diff_amd']
import numpy as np, torch
import mudiff_hip
B, rounds = int(sys.argv[1]), int(sys.argv[2])
libs = []
for p in sys.argv[3:]:
    lib = C.CDLL(os.path.abspath(p))
    for name in ('mud_attention', 'mud_attention_ws_bytes'):
        res, args = mudiff_hip._SIGNATURES[name]
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    libs.append(lib)
names = [os.path.basename(p).replace('lib_', '').replace('.so', '') for p in sys.argv[3:]]
dev = 'cuda:0'
N, Cc = 4096, 256
qkv = torch.randn(B, N, 3 * Cc, device=dev)
out = torch.empty(B, N, Cc, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
ws = torch.empty(max(1, max(lib.mud_attention_ws_bytes(B, N, Cc) for lib in libs)), device=dev, dtype=torch.uint8)
times = [[] for _ in libs]
outs = []
for rd in range(rounds + 1):
    for i, lib in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            assert lib.mud_attention(P(qkv), B, N, Cc, 3 * Cc, Cc ** -0.5, P(out), Cc, P(ws), stream) == 0
        e1.record(); torch.cuda.synchronize()
        if rd:
            times[i].append(e0.elapsed_time(e1) / 5 * 1e3)
        else:
            outs.append(out.clone())
med = [float(np.median(t)) for t in times]
fl = 4.0 * B * N * N * Cc
print(f'attention B={B} N={N} C={Cc}: ' + '  '.join(f'{n} {m:8.1f}us ({fl / m / 1e6:5.0f} TF, {m / med[0]:.3f})' for n, m in zip(names, med)),
      '| max|a - b|', ' '.join(f'{float((o - outs[0]).abs().max()):.1e}' for o in outs[1:]))
