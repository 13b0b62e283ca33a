"""Micro-benchmark of the path's fp32 MFMA GEMM through tmdnet_debug_gemm (developer tool)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import _C
L = _C.lib()
P = 193710
shapes = [(P, 384, 256, 1), (P, 256, 128, 1), (P, 128, 32, 1), (P, 256, 384, 0), (P, 128, 256, 0), (P, 32, 128, 0), (P, 384, 32, 0),
          (16384 * 9, 128, 128, 0), (16384, 256, 128, 1), (16384, 384, 256, 1), (16384, 128, 384, 1), (100, 128, 128, 0)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, act) in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    Cc = torch.empty(M, N, device="cuda")
    Wh = W.cpu().contiguous()
    nsb = L.tmdnet_debug_split_weight(C.c_void_p(Wh.data_ptr()), N, K, None)
    img = torch.empty(nsb, dtype=torch.int16)
    L.tmdnet_debug_split_weight(C.c_void_p(Wh.data_ptr()), N, K, C.c_void_p(img.data_ptr()))
    Wsb = img.cuda()
    use_sb = os.environ.get("SB", "1") == "1"
    args = (s, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(Cc.data_ptr()), M, N, K, act,
            C.c_void_p(Wsb.data_ptr()) if use_sb else None)
    for _ in range(3): L.tmdnet_debug_gemm(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.tmdnet_debug_gemm(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * M * N * K / 1e9
    gb = 4.0 * (M * K + N * K + M * N) / 1e9
    ref = A[:4096].double() @ W.double().t() + b.double()
    if act: ref = torch.nn.functional.silu(ref)
    err = (Cc[:4096].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M={M:7d} N={N:4d} K={K:4d} act={act}: {us:8.1f} us  {gf/us*1e3:6.1f} TF/s  {gb/us*1e3:6.2f} TB/s  relerr {err:.1e}", flush=True)
