"""configs[4] layer shapes on each tile shape of the streamed-weight conv kernel (tools build: ap_debug_conv_path(4..7) forces
128x128 / 128x64 / 64x64 / 64x128 for every layer, 8 = the product's dispatch): ms and fraction of the fp32 MFMA peak per shape.
python tools/sweep_conv_tiles.py"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib
import ctypes as C, torch
from audiopure_amd import _native as N
dev = torch.device("cuda:0"); lib = N.lib()
dl = C.CDLL(N.LIB_PATH); dl.ap_debug_conv_path.argtypes = [C.c_int]
N.use_conv_workspace(dev)
shapes = [(256, 128, 32, 32, 128, 3), (256, 256, 16, 16, 256, 3), (256, 512, 16, 16, 256, 3), (256, 256, 32, 32, 128, 3), (256, 256, 8, 8, 256, 3), (256, 512, 8, 8, 256, 3),
          (256, 256, 4, 4, 256, 3), (256, 512, 4, 4, 256, 3), (256, 256, 16, 16, 768, 1), (256, 256, 16, 16, 256, 1), (256, 256, 8, 8, 768, 1), (256, 256, 8, 8, 256, 1), (256, 512, 8, 8, 256, 1),
          (256, 256, 32, 32, 128, 1), (256, 512, 16, 16, 256, 1), (256, 256, 4, 4, 768, 1), (256, 256, 4, 4, 256, 1)]
for (B, Cin, H, W, Cout, k) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    res = torch.randn(B, Cout, H, W, device=dev)
    wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, k, k, 1), device=dev)
    N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, k, k, 1, N.stream()))
    out = torch.empty(B, Cout, H, W, device=dev)
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    tiles = ((B * H * W + 127) // 128) * ((Cout + 127) // 128)
    line = f"{Cin:4d}->{Cout:4d} {H:2d}x{W:2d} k{k} tiles128 {tiles:5d}:"
    for thr in (8, 4, 5, 6, 7):
        dl.ap_debug_conv_path(thr)
        def launch(n):
            for _ in range(n):
                N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(bias), N.ptr(res), N.ptr(out), B, Cin, H, W, Cout, k, k, 1, k // 2, 1, 0, Cin, 0, N.stream()))
        launch(5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(50); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        line += "   %7s %.4f %.3f" % ({8: 'product', 4: '128x128', 5: '128x64', 6: '64x64', 7: '64x128'}[thr], ms, fl / ms / 1e9 / 157.3)
    print(line)
dl.ap_debug_conv_path(8)
