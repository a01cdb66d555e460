"""ORACLE — test infrastructure only.  Interprets a lowered ConvNet plan (audiopure_amd.convnet.Plan) with plain
PyTorch-CPU ops, so the lowering (ATen tape -> fused plan) can be checked against the original nn.Module without a
GPU, and the HIP executor can be checked against the same plan on the GPU box.  The reference semantics are those of
the module itself: audio_models/ConvNets_SpeechCommands/models/*.py forward()."""
import torch
import torch.nn.functional as F


def run_plan_torch(plan, x: torch.Tensor) -> torch.Tensor:
    B = x.shape[0]
    bufs = {plan.input.buf: x.float()}

    def buf(v):
        if v.buf not in bufs:
            C, H, W = plan.buf_shape[v.buf]
            bufs[v.buf] = torch.full((B, C, H, W), float("nan"))
        return bufs[v.buf]

    def rd(v):
        return buf(v)[:, v.coff:v.coff + v.C].reshape(B, v.C, v.H, v.W)

    def wr(v, t):
        buf(v)[:, v.coff:v.coff + v.C] = t.reshape(B, v.C, *buf(v).shape[2:])

    Wt = plan.weights
    for s in plan.steps:
        p = s.p
        if s.kind == "conv":
            w = Wt[p["wk"]]
            if p["sk"]:
                w = w * Wt[p["sk"]].view(-1, 1, 1, 1)
            y = F.conv2d(rd(s.ins[0]), w, Wt[p["bk"]] if p["bk"] else None, stride=p["stride"], padding=p["pad"],
                         groups=p["groups"])
            if p["res"] is not None:
                y = y + rd(p["res"])
            wr(s.out, F.relu(y) if p["relu"] else y)
        elif s.kind == "affine":
            y = rd(s.ins[0])
            if p["sk"]:
                y = y * Wt[p["sk"]].view(1, -1, 1, 1) + Wt[p["hk"]].view(1, -1, 1, 1)
            wr(s.out, F.relu(y) if p["relu"] else y)
        elif s.kind == "add":
            y = rd(s.ins[0]) + rd(s.ins[1])
            wr(s.out, F.relu(y) if p["relu"] else y)
        elif s.kind == "copy":
            wr(s.out, rd(s.ins[0]))
        elif s.kind == "pool":
            f = F.max_pool2d if p["is_max"] else F.avg_pool2d
            wr(s.out, f(rd(s.ins[0]), p["k"], p["stride"], p["pad"]))
    o = plan.output
    y = rd(o)
    return y.reshape(B, -1) if (o.H == 1 and o.W == 1) else y
