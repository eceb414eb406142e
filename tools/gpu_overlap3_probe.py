"""GPU probe (not a pytest): would carrying a layer's grouped weight gradients (gemm_f32_x3_group) on a second stream,
beside the NEXT layer's backward chain (dgrad GEMMs, LayerNorm backward, attention backward), shorten the step?
Shapes of one C2 transformer layer (rows = 16384, H = 512, FFN 2048, 8 heads x 64).  Serial vs two streams, both captured
into hipGraphs of 4 layers each (launch overhead out of the picture)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

dev = "cuda"
R, H, F, nh = 16384, 512, 2048, 8
B, T = 256, 64
torch.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev)
# saved activations / gradients of a layer
x_ln1, x_ln2, attn_o, hmid = rn(R, H), rn(R, H), rn(R, H), rn(R, F)
d_out, d_h, d_ao, d_qkv = rn(R, H), rn(R, F), rn(R, H), rn(R, 3 * H)
Wqkv, Wo, W1, W2 = rn(3 * H, H), rn(H, H), rn(F, H), rn(H, F)
gW = [torch.empty_like(w) for w in (Wqkv, Wo, W1, W2)]
gb = [torch.empty(w.shape[0], device=dev) for w in (Wqkv, Wo, W1, W2)]
ws1, ws2 = torch.empty(48 << 20, device=dev), torch.empty(48 << 20, device=dev)
qkv = rn(R, 3 * H)
o, lse = rn(R, H), rn(B * nh * T)
dqkv, dsum = torch.empty_like(qkv), torch.empty(B * nh * T, device=dev)
valid = torch.ones(B, T, dtype=torch.bool, device=dev)
rng = torch.tensor([7, 5], device=dev, dtype=torch.int64)
lnw, mean, rstd = rn(H), rn(R), rn(R).abs() + 0.5
dx, dlw, dlb = torch.empty(R, H, device=dev), torch.empty(H, device=dev), torch.empty(H, device=dev)
t1, t2 = torch.empty(R, F, device=dev), torch.empty(R, H, device=dev)
lnws = torch.empty(4 << 20, device=dev)


def wgrads(ws):
    items = [(d_qkv, 3 * H, x_ln1, H, R, gW[0], gb[0]), (d_ao, H, attn_o, H, R, gW[1], gb[1]),
             (d_h, F, x_ln2, H, R, gW[2], gb[2]), (d_out, H, hmid, F, R, gW[3], gb[3])]
    L.wgrad_group(items, ws)


def chain(ws):
    """the backward chain of one layer without its weight gradients"""
    L.gemm(1, 0, R, F, H, d_out, H, W2, F, t1, F, mode=L.EPI_MUL_POSMASK, aux_in=hmid, ldaux=F, p=0.1, ws=ws)      # FFN2 dgrad
    L.gemm(1, 0, R, H, F, t1, F, W1, H, t2, H, ws=ws)                                                             # FFN1 dgrad
    L.call("vqh_layernorm_bwd", t2, H, x_ln2, H, lnw, mean, rstd, dx, H, 0, dlw, dlb, 0.0, R, H, None, None, 0, 0.0, lnws, lnws.numel())
    L.gemm(1, 0, R, H, H, dx, H, Wo, H, t2, H, ws=ws)                                                             # out-proj dgrad
    L.call("vqh_attn_bwd", qkv, 3 * H, qkv[:, H:], 3 * H, qkv[:, 2 * H:], 3 * H, o, H, lse, t2, H, dsum, dqkv, 3 * H, dqkv[:, H:], 3 * H,
           dqkv[:, 2 * H:], 3 * H, valid, B, nh, T, T, 64, 0, rng, 5, 0.1)
    L.gemm(1, 0, R, H, 3 * H, dqkv, 3 * H, Wqkv, H, t2, H, ws=ws)                                                 # QKV dgrad
    L.call("vqh_layernorm_bwd", t2, H, x_ln1, H, lnw, mean, rstd, dx, H, 0, dlw, dlb, 0.0, R, H, None, None, 0, 0.0, lnws, lnws.numel())


s2 = torch.cuda.Stream()


def serial(n=4):
    for _ in range(n):
        chain(ws1)
        wgrads(ws1)


def forked(n=4):
    """wgrads of layer l on the side stream while the main stream runs the chain of layer l-1"""
    ev_join = None
    for _ in range(n):
        chain(ws1)
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            wgrads(ws2)
            ev_join = torch.cuda.Event(); ev_join.record()
    torch.cuda.current_stream().wait_event(ev_join)


def timed_graph(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3 / 4


for rnd in range(2):
    a, b = timed_graph(serial), timed_graph(forked)
    only_chain = timed_graph(lambda n=4: [chain(ws1) for _ in range(n)])
    only_w = timed_graph(lambda n=4: [wgrads(ws1) for _ in range(n)])
    print(f"per layer: chain alone {only_chain:.1f} us, wgrad group alone {only_w:.1f} us, serial {a:.1f} us, two streams {b:.1f} us "
          f"({(1 - b / a) * 100:+.1f} %)")
