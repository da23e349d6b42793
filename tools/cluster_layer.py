#!/usr/bin/env python3
"""Drives tools/cluster_layer.hip (the MOCK of DESIGN.md section 7's batch-element-persistent encoder stack): builds it on the box,
checks every stored tensor of a 1- and a 6-layer run against plain torch (fp32 arithmetic on the same bf16 weights, bf16 rounding
where the kernel stores), then times it against the library's encoder stack forward on the same shape (64 sentences x 128 tokens,
d = 512, 8 heads, ff = 2048, no dropout, no padding).  GPU only.   python tools/cluster_layer.py [layers]"""
import ctypes, os, subprocess, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, T, D, H, FF = 64, 128, 512, 8, 2048
dev = torch.device("cuda")


def build():
    so = "/tmp/libcluster_layer.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-DXCD_LOCAL=%d" % int(os.environ.get("XCD_LOCAL", "0")),
                           os.path.join(ROOT, "tools", "cluster_layer.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.cl_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    lib.cl_run.restype = ctypes.c_int
    return lib


def make_layer(g):
    def w(*shape, s=0.03):
        return (torch.randn(*shape, generator=g, device=dev) * s).bfloat16()
    return dict(wqkv=w(3 * D, D), bqkv=w(3 * D), wo=w(D, D), bo=w(D), g1=(1 + w(D)).bfloat16(), b1=w(D),
                w1=w(FF, D), bf1=w(FF), w2=w(D, FF), bf2=w(D), g2=(1 + w(D)).bfloat16(), b2=w(D))


def make_bufs():
    z = lambda *s, dt=torch.bfloat16: torch.zeros(*s, device=dev, dtype=dt)
    return dict(qkv=z(B * T, 3 * D), ctx=z(B * T, D), y1=z(B * T, D), pre=z(B * T, FF), h=z(B * T, FF), y2=z(B * T, D),
                st1=z(4, B * T, 2, dt=torch.float32), st2=z(4, B * T, 2, dt=torch.float32))


WKEYS = ["wqkv", "bqkv", "wo", "bo", "g1", "b1", "w1", "bf1", "w2", "bf2", "g2", "b2"]
BKEYS = ["qkv", "ctx", "y1", "pre", "h", "y2", "st1", "st2"]


def reference_layer(x, w):
    """torch restatement with the kernel's rounding points; x bf16 [B*T, D]"""
    f = lambda t: t.float()
    qkv = (f(x) @ f(w["wqkv"]).t() + f(w["bqkv"])).bfloat16()
    q, k, v = [f(t).view(B, T, H, 64).transpose(1, 2) for t in qkv.split(D, dim=1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = torch.softmax(s, dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B * T, D).bfloat16()
    a = f(ctx) @ f(w["wo"]).t() + f(w["bo"]) + f(x)
    y1 = (F.layer_norm(a, (D,), eps=1e-12) * f(w["g1"]) + f(w["b1"])).bfloat16()
    z = f(y1) @ f(w["w1"]).t() + f(w["bf1"])
    pre, h = z.bfloat16(), F.gelu(z).bfloat16()
    o = f(h) @ f(w["w2"]).t() + f(w["bf2"]) + f(y1)
    y2 = (F.layer_norm(o, (D,), eps=1e-12) * f(w["g2"]) + f(w["b2"])).bfloat16()
    return dict(qkv=qkv, ctx=ctx, y1=y1, pre=pre, h=h, y2=y2)


class Runner:
    def __init__(self, lib, layers):
        g = torch.Generator(device=dev).manual_seed(7)
        self.lib, self.layers = lib, layers
        self.x0 = torch.randn(B * T, D, generator=g, device=dev).bfloat16()
        self.w = [make_layer(g) for _ in range(layers)]
        self.bufs = [make_bufs() for _ in range(layers)]
        ptrs = [self.x0.data_ptr()]
        for l in range(layers):
            ptrs += [self.w[l][k].data_ptr() for k in WKEYS] + [self.bufs[l][k].data_ptr() for k in BKEYS]
        self.ptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        self.flags = torch.zeros(256 * 32, device=dev, dtype=torch.int32)
        self.status = torch.zeros(16, device=dev, dtype=torch.int32)
        self.epoch = 1
        self.trace = None

    def run(self):
        rc = self.lib.cl_run(self.ptrs, self.layers, B, self.flags.data_ptr(), self.status.data_ptr(), self.epoch,
                             torch.cuda.current_stream().cuda_stream, self.trace.data_ptr() if self.trace is not None else None)
        assert rc == 0, rc
        self.epoch += 6 * self.layers + 2


def check(r):
    r.run()
    torch.cuda.synchronize()
    st = int(r.status[0])
    assert st == 0, "a bounded wait ran out: status 0x%x" % st
    x = r.x0
    worst = {}
    for l in range(r.layers):
        ref = reference_layer(x, r.w[l])
        for k, v in ref.items():
            got = r.bufs[l][k].float()
            err = float((got - v.float()).abs().max() / v.float().abs().max())
            worst[k] = max(worst.get(k, 0.0), err)
        x = r.bufs[l]["y2"]   # layer by layer against the kernel's own input: rounding does not compound in the check
    print("layers=%d  max|got-ref|/max|ref| per stored tensor: %s" % (r.layers, "  ".join("%s %.1e" % kv for kv in worst.items())))
    assert max(worst.values()) < 2e-2, worst


def time_it(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def library_encoder_us():
    from bench import CONFIGS, build_model
    c = CONFIGS["c1"]
    model = build_model(c, torch.bfloat16, dev).eval()   # dropout off, parameters require grad: the forward saves for the backward
    ids = torch.randint(5, c["V"], (B, T), device=dev)
    mask = torch.ones(B, T, dtype=torch.bool, device=dev)
    langs = torch.zeros(B, T, dtype=torch.long, device=dev)
    def fwd():
        return model.encode(ids, mask, langs)[0]
    return time_it(fwd)


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    lib = build()
    check(Runner(lib, 1))
    r = Runner(lib, layers)
    check(r)
    us1 = time_it(Runner(lib, 1).run)
    usL = time_it(r.run)
    assert int(r.status[0]) == 0
    print("cluster-persistent mock: 1 layer %.1f us per launch; %d layers %.1f us per launch = %.1f us per layer" % (us1, layers, usL, usL / layers))
    r.trace = torch.zeros(256, 8, 16, device=dev, dtype=torch.int64)
    r.run()
    torch.cuda.synchronize()
    tr = r.trace[:, :layers, :12].double() / 100.0   # us
    names = ["wait for input", "q|k|v product", "attention + publish", "wait for contexts", "projection", "LayerNorm (exchange inside) + publish",
             "wait for y1", "FFN-up + GELU + publish", "wait for h", "FFN-down", "LayerNorm (exchange inside) + publish"]
    d = (tr[:, :, 1:] - tr[:, :, :-1])
    print("phase lengths, mean over the 256 workgroups and layers 1.. (us); layer 0 in brackets:")
    for k, nm in enumerate(names):
        print("  %-42s %6.2f   [%6.2f]" % (nm, float(d[:, 1:, k].mean()) if layers > 1 else float("nan"), float(d[:, 0, k].mean())))
    print("  %-42s %6.2f" % ("layer (stamp 0 -> 11)", float((tr[:, :, 11] - tr[:, :, 0]).mean())))
    print("  launch span (first stamp -> last stamp over all workgroups): %.1f us" % float(tr[:, layers - 1, 11].max() - tr[:, 0, 0].min()))
    r.trace = None
    lib_us = library_encoder_us()
    print("library encoder stack forward (embedding + %d layers, eval mode with saved activations): %.1f us = %.1f us per layer" % (6, lib_us, lib_us / 6))


if __name__ == "__main__":
    main()
