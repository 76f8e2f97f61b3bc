"""torch's own kernels as victims of the cross-wave packed-fp32 disturbance (profiles/ub/pk_hazard.hip is the stand-alone form).

The in-tree kernels are built without packed fp32; torch's element-wise / fused-Adam kernels and hipBLASLt are not ours to rebuild and
run on the other streams of a training step.  This probe runs >= 10^4 launches of each victim kind while `gather_conv_split_kernel`
(256 -> 256, 3x3, 128^2, the aggressor of every earlier probe) occupies a side stream, and compares every result bit for bit with the
same launch made with the device otherwise idle.

    python profiles/pk_hazard_torch_probe.py [launches_per_kind]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc  # noqa: E402

dev = torch.device("cuda:0")
N_LAUNCH = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
BATCH = 250
torch.manual_seed(5)
side = torch.cuda.Stream()
conv_x = torch.randn(1, 256, 128, 128, device=dev)
conv_w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
apply = agc._Conv.apply


def aggressor(n):
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(n):
            apply(conv_x, conv_w, None, None, agc.AG_CONV, 1, 1, 1.0)


def make_victims():
    n = 1 << 18                                    # 1 MB tensors: a launch is a few microseconds, hundreds fit under one aggressor burst
    a, b, c = (torch.randn(n, device=dev) for _ in range(3))
    a4 = torch.randn(64, 4096, device=dev)
    victims = {
        "add (vectorized_elementwise)": lambda: torch.add(a, b, alpha=0.37),
        "mul": lambda: torch.mul(a, b),
        "addcmul": lambda: torch.addcmul(a, b, c, value=0.5),
        "lerp": lambda: torch.lerp(a, b, 0.3),
        "leaky_relu": lambda: torch.nn.functional.leaky_relu(a, 0.2),
        "sum(dim) reduction": lambda: a4.sum(dim=1),
        "softmax": lambda: torch.softmax(a4, dim=1),
        "mm 512x512 (hipBLASLt/rocBLAS)": None,
        "fused Adam step": None,
    }
    m1, m2 = torch.randn(512, 512, device=dev), torch.randn(512, 512, device=dev)
    victims["mm 512x512 (hipBLASLt/rocBLAS)"] = lambda: m1 @ m2
    return victims


def run_kind(name, fn):
    bad_launches = bad_elems = launches = 0
    with torch.no_grad():
        ref = fn().clone()
        torch.cuda.synchronize()
        while launches < N_LAUNCH:
            aggressor(60)
            outs = [fn() for _ in range(BATCH)]
            torch.cuda.synchronize()
            for o in outs:
                d = int((o.view(torch.int32) != ref.view(torch.int32)).sum())
                bad_elems += d
                bad_launches += d > 0
            launches += BATCH
    return {"launches": launches, "launches_with_a_wrong_bit": bad_launches, "wrong_elements": bad_elems, "elements_per_launch": ref.numel()}


def run_adam():
    """Fused Adam over 40 tensors (2.6 M parameters), N steps under the aggressor against the same N steps on an idle device."""
    shapes = [(512, 512, 3, 3)] + [(256, 256)] * 8 + [(512,)] * 31
    g = torch.Generator(device=dev).manual_seed(3)
    init = [torch.randn(s, device=dev, generator=g) for s in shapes]
    grads = [[torch.randn(s, device=dev, generator=g) * 0.01 for s in shapes] for _ in range(4)]

    def steps(n, disturbed):
        params = [p.clone().requires_grad_(True) for p in init]
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)
        done = 0
        while done < n:
            if disturbed:
                aggressor(60)
            for _ in range(min(BATCH, n - done)):
                for p, gr in zip(params, grads[done % 4]):
                    p.grad = gr
                opt.step()
                done += 1
            torch.cuda.synchronize()
        return [p.detach() for p in params]

    n = max(1000, N_LAUNCH // 10)
    ref, got = steps(n, False), steps(n, True)
    wrong = sum(int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(ref, got))
    return {"launches": n, "wrong_elements_after_all_steps": wrong, "elements": sum(p.numel() for p in ref)}


def main():
    agc.set_math("split_bf16")
    out = {"aggressor": "gather_conv_split_kernel 256->256 3x3 @128^2 on a side stream", "device": torch.cuda.get_device_name(0), "kinds": {}}
    for name, fn in make_victims().items():
        out["kinds"][name] = run_adam() if fn is None and "Adam" in name else run_kind(name, fn)
        print(name, out["kinds"][name], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
