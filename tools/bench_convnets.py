import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from synth_convnets import CifarResNeXt, synth_init, vgg19_bn
from audiopure_amd.convnet import NativeConvNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 1, 32, 32, device=dev)
for name, make, gflop in (("resnext29_8_64", lambda: CifarResNeXt(10), 10.77), ("vgg19_bn", lambda: vgg19_bn(10, 1), 0.83)):
    net = NativeConvNet(synth_init(make(), 0)).eval()
    for _ in range(2): net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: B={B} {dt*1e3:.2f} ms  {B/dt:.0f} samples/s  {gflop*B/dt/1e3:.1f} TFLOP/s")
