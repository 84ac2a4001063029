import os, sys, torch
sys.path.insert(0, "/root/repo")
from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net
dev = torch.device("cuda:0")
H, W = 1152, 1600
layers = [("conv0.0 3->8", 1, 3, 8, 3, H, W), ("conv0.1 8->8", 1, 8, 8, 3, H, W), ("16->16", 1, 16, 16, 3, H // 2, W // 2),
          ("out3 64->16", 1, 64, 16, 3, H // 2, W // 2), ("out4 64->8", 1, 64, 8, 3, H, W), ("reg conv0 4->8", 4, 4, 8, 3, H, W),
          ("reg conv2 48->16", 4, 48, 16, 3, H // 2, W // 2), ("prob 8->1", 4, 8, 1, 1, H, W), ("mono 16->1", 1, 16, 1, 3, H, W),
          ("mono cb 16->8", 1, 16, 8, 3, H // 2, W // 2), ("reg s2 conv0 8->8 (8 slices 288x400)", 8, 8, 8, 3, H // 4, W // 4)]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, N, ci, co, k, h, w in layers:
    x = torch.randn(N, ci, h, w, device=dev); wt = torch.randn(co, ci, k, k, device=dev); gy = torch.randn(N, co, h, w, device=dev)
    cb = lambda mask: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [k // 2] * 2, [1, 1], False, [0, 0], 1, mask)
    m_d, m_w = t(lambda: cb([True, False, False])), t(lambda: cb([False, True, False]))
    lib = mvs4net._lib.load()
    wtd = wt.flip(2, 3).permute(0, 2, 3, 1).contiguous(); dx = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    d_d = t(lambda: mvs4net._lib.call("mvster_conv_direct_fwd", gy.data_ptr(), wtd.data_ptr(), None, dx.data_ptr(), N, co, ci, h, w, k, st)) if lib.mvster_conv_direct_supported(co, ci, k) else float("nan")
    need = lib.mvster_conv_direct_wgrad_workspace(N, ci, co, h, w, k); ws = torch.empty(need, device=dev); dw = torch.empty_like(wt)
    d_w = t(lambda: mvs4net._lib.call("mvster_conv_direct_wgrad", x.data_ptr(), gy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), need, N, ci, co, h, w, k, st))
    print("%-40s MIOpen dgrad %.3f wgrad %.3f | direct dgrad %.3f wgrad %.3f" % (name, m_d, m_w, d_d, d_w), flush=True)
