# -*- coding: utf-8 -*-
"""
StepEngine: the hand-scheduled forward / backward / optimizer step of the VQ-VAE hot path, written
against the C ABI of libvqvae_hip.so (include/vqvae_hip.h).  No torch.autograd, no torch compute
kernels: torch only provides device buffers, the current stream and (for N>1 ranks) RCCL through
torch.distributed.  Every launch goes to the current stream, so a whole step can be captured into
a hipGraph (torch.cuda.graph) and replayed.

Mirrors, op for op, the reference call stack:
  VQVAE.forward /root/reference/models/vq_vae.py:767-901  -> encode :639-660, LatentTokenizer :310-322,
  to_code :736-743, VectorQuantizerEMA.forward :170-283, decode :745-765, loss_function :1097-1388,
  then backward + clip_grad_norm_ + AdamW (/root/reference/experiment.py:169-197, run.py:191-197).

Memory layout in HBM (all fp32, row-major):
  * parameters / gradients / Adam moments: ONE flat buffer each (every tensor padded to 16 B), the
    nn.Parameter objects are views -> grad all-reduce, global-norm clip and AdamW are single launches
  * activations: [rows, features] matrices with rows = B*L (sequence side) or B*N (latent side);
    attention reads heads in place from the packed QKV projection (no head transposes)
  * residual-stream gradient: one [rows, H] buffer per stack, updated in place layer by layer
"""
import collections
import gc
import os
import weakref

import torch

from . import lib as L
from .lib import call

SS_LAYERS = 2        # models/vq_vae.py:473
LN_EPS = 1e-5
VQ_EPS = 1e-5
WS_FLOATS = 48 * 1024 * 1024

def bwd_phases(num_layers):
    """Backward phases in completion order: the parameter-name prefixes whose gradients are final when the phase ends
    (see StepEngine._flatten / backward_gen).  One phase per decoder layer, the tokenizer, the SS encoder + fusion MLP, and
    one per geometry-encoder layer: each phase's gradients are one contiguous all-reduce bucket that travels while the
    next phase computes, and the only exposed bucket is the last one -- geometry layer 0 + input_proj, 7 % of the bytes
    at C2 (round 1 used 4 coarse phases and left 29 % exposed)."""
    nl = int(num_layers)
    dec = []
    for i in reversed(range(nl)):
        pre = (f"decoder.layers.{i}.",)
        if i == nl - 1:
            pre += ("head_xyz.", "head_ss.")
        if i == 0:
            pre += ("query_embed.", "mem_ln.", "from_code.")
        dec.append(pre)
    if not dec:
        dec = [("head_xyz.", "head_ss.", "query_embed.", "mem_ln.", "from_code.")]
    geo = []
    for i in reversed(range(nl)):
        pre = (f"encoder.layers.{i}.",)
        if i == nl - 1:
            pre += ("ln_geo.", "enc_ln.")
        if i == 0:
            pre += ("input_proj.",)
        geo.append(pre)
    if not geo:
        geo = [("ln_geo.", "enc_ln.", "input_proj.")]
    return tuple(dec) + (("tokenizer.", "to_code."), ("fuse_mlp.", "ln_ss.", "ss_encoder.", "ss_input_proj.")) + tuple(geo)


def bucket_length(L0, granule, max_seq_len):
    """Padded sequence length of the fused step: the next multiple of `granule`, capped at max_seq_len (pos_enc and
    query_embed have max_seq_len rows: models/vq_vae.py:639-650, :750-751).  L <= 350 at granule 32 -> 32, 64, ..., 320, 350:
    11 shapes instead of one per distinct L_max, and B*L a multiple of the 256-row GEMM tile for every even batch >= 8."""
    L0, g = int(L0), int(granule)
    if g <= 1:
        return L0
    return int(min(-(-L0 // g) * g, max(int(max_seq_len), L0)))


def _bwd_phase(name, phases):
    for i, prefixes in enumerate(phases):
        if name.startswith(prefixes):
            return i
    raise KeyError(f"parameter {name!r} is not assigned to a backward phase")


METRIC_KEYS = ["loss", "Reconstruction_Loss_XYZ", "XYZ_MSE_Raw", "XYZ_MSE_Aligned", "Reconstruction_Loss_SS",
               "SS_Accuracy", "VQ_Loss", "Geom_BondLength_Loss", "Geom_BondAngle_Loss", "Geom_Direction_Loss",
               "Geom_Dihedral_Loss", "Geom_Loss", "SS_TV", "Usage_Reg", "XYZ_TV2", "VQ_Perplexity", "VQ_DeadRatio",
               "RMSD_Raw", "RMSD_Aligned", "Geom_LocalPDM", "Geom_WinKabsch", "Frenet_Kappa", "Frenet_Tau",
               "Geom_LongRangePDM"]
OPTIONAL_METRICS = {"Geom_LocalPDM": "pdm_weight", "Geom_WinKabsch": "win_kabsch_weight", "Frenet_Kappa": "kappa_weight",
                    "Frenet_Tau": "tau_weight", "Geom_LongRangePDM": "lr_pdm_weight"}


class _Arena:
    """Every device buffer the step needs at ONE batch shape, plus the hipGraphs captured over them.  A captured graph
    addresses its buffers by raw pointer, so buffers and graphs share one lifetime: an arena is only ever dropped whole
    (LRU, StepEngine.use_arena).  Inside an arena a buffer is keyed by (name, shape, dtype) and never re-allocated."""
    __slots__ = ("key", "pool", "buf", "graphs", "seen", "nbytes")

    def __init__(self, key):
        self.key = key
        self.pool = {}                              # (name, shape, dtype) -> tensor, alive as long as the arena
        self.buf = {}                               # name -> the tensor most recently handed out under that name
        self.graphs = collections.OrderedDict()     # step key -> tuple of captured graphs (LRU, MAX_GRAPHS_PER_ARENA)
        self.seen = {}
        self.nbytes = 0

    def release(self):
        self.graphs.clear()                         # graphs first: they reference the buffers below
        self.seen.clear()
        self.buf.clear()
        self.pool.clear()
        self.nbytes = 0


MAX_GRAPHS_PER_ARENA = 4
HYPER_SLOTS = 4          # pinned host copies of the per-step scalars in flight = how many steps the host may run ahead


class StepEngine:
    def __init__(self, model, seed=None):
        L.require_gpu()
        # weak: the model owns the engine (VQVAE._eng); a strong back reference would make the pair a reference cycle, whose
        # hipGraphs are then destroyed whenever the cycle collector happens to run -- e.g. in the middle of ANOTHER
        # engine's stream capture, which aborts the process
        self._model_ref = weakref.ref(model)
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise L.VqhError("StepEngine: move the model to the GPU first (no CPU path exists)")
        self.H = model.hidden_dim
        self.D = model.code_dim
        self.nh = model.num_heads
        self.tnh = model.tokenizer_heads
        self.N = model.latent_n_tokens
        # batch-shape arenas (see _Arena): real data has a different L_max per batch (dataset.py:30-49) and a validation
        # pass between train epochs (experiment.py:478-479); each shape keeps its own buffers + graphs, LRU-bounded
        self.arenas = collections.OrderedDict()
        self.max_arenas = int(os.environ.get("VQH_MAX_ARENAS", "16"))     # > the 11 length buckets of L <= 350 (bucket_len)
        # Length bucketing of the fused train / eval step (real data: pad_collate pads every batch to its OWN L_max,
        # /root/reference/dataset.py:30-49): L_max is padded up to a multiple of `len_bucket` (mask False on the tail, every
        # kernel is mask-aware), so that few arenas / graphs exist and B*L stays a multiple of the 256-row GEMM tile.
        self.len_bucket = int(os.environ.get("VQH_LEN_BUCKET", "32"))
        self.max_arena_bytes = int(float(os.environ.get("VQH_ARENA_GIB", "0")) * (1 << 30)) or \
            int(0.6 * torch.cuda.get_device_properties(self.dev).total_memory)
        self.arena = None
        self.buf = None
        self.use_arena(("init",))
        self.ws = torch.empty(WS_FLOATS, device=self.dev, dtype=torch.float32)
        # dropout counter-hash state [seed, step]; the seed follows torch.manual_seed (exp_params.manual_seed, run.py:119-121)
        if seed is None:
            seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        self.rng = torch.tensor([int(seed), 0], device=self.dev, dtype=torch.int64)
        self.drop_scale = 1.0                      # 0.0 disables every dropout site (parity runs)
        self._sites = {}
        self._flatten()
        self.hyper = torch.zeros(10, device=self.dev, dtype=torch.float32)    # 9 optimizer scalars + the batch's own L_max
        self._hyper_host, self._hyper_ev = None, None
        self.metrics_acc = torch.zeros(len(METRIC_KEYS), device=self.dev, dtype=torch.float32)
        self._pending_ema = None
        self.fwd_id = 0                            # incremented by every forward: ties loss_function/backward to it
        self.last_step_mode = None                 # "eager" | "capture" | "graph": how the last train_step ran
        self.norm = torch.zeros(2, device=self.dev, dtype=torch.float32)
        self.norm_ws = torch.empty(1024, device=self.dev, dtype=torch.float64)
        self.metrics = torch.zeros(len(METRIC_KEYS), device=self.dev, dtype=torch.float32)
        self.vq_stats = torch.zeros(2, device=self.dev, dtype=torch.float32)
        self.opt_step = 0
        self.train = True
        self.ctx = None
        self.defer_ema = False
        # first decoder / tokenizer layer: project the batch-invariant queries once instead of per sample
        self.share_layer0 = os.environ.get("VQH_SHARE_LAYER0", "1") != "0"
        # dropout backward of a residual branch written by the LayerNorm backward that produces its input (ln_bwd emit)
        self.fold_dropout_bwd = os.environ.get("VQH_FOLD_DROPOUT_BWD", "1") != "0"
        # weight gradients of a transformer layer are queued and run as ONE grouped launch when the layer's backward is
        # done (vqh_gemm_wgrad_group): long K ranges per workgroup and few split-K slabs instead of 4-7 short launches
        self.group_wgrad = os.environ.get("VQH_GROUP_WGRAD", "1") != "0"
        self._wq = []
        self._dy_slot = 0
        # EXPERIMENT, off by default (VQH_OVERLAP_WGRAD=1): the grouped weight-gradient launch of a layer on a SECOND stream
        # while the main stream goes on with the next layer's backward (nothing reads a weight gradient before the clip), so
        # that the prologue / epilogue / drain gaps of one kernel are filled with workgroups of the other.  Measured on C2:
        # 27.04 ms per step against 26.64 ms in order -- both kernels want a whole CU's LDS, so they only alternate, and two
        # operand working sets share each XCD's L2.  Results are identical (same kernels, same order per gradient).
        # Single process only: with several ranks a phase's gradients must be final at its bucket's all-reduce, and a graph
        # segment has to rejoin its forked streams.
        self.overlap_wgrad = self.group_wgrad and os.environ.get("VQH_OVERLAP_WGRAD", "0") == "1"
        self._side = None                     # the second stream
        self._ws_side = None                  # its own split-K workspace
        self._side_busy = False
        self._par = 0                         # parity of the operand buffers handed out by TW()
        self._buf_ev = {}                     # storage pointer of an operand buffer -> event of the side launch that reads it

    @property
    def m(self):
        model = self._model_ref()
        if model is None:
            raise L.VqhError("StepEngine: its model has been deleted")
        return model

    # ------------------------------------------------------------------ parameters
    def _flatten(self):
        """Re-home every parameter into one flat buffer (views), plus flat grad / Adam moments."""
        named = list(self.m.named_parameters())
        # Layout = the order in which backward completes the gradients, so that each phase's gradients are one
        # contiguous bucket whose all-reduce can start while the next phase still computes:
        #   decoder layers n-1..0 | tokenizer | SS encoder + fuse | geometry layers n-1..0 | EMA statistics   (bwd_phases)
        self.phases = bwd_phases(self.m.num_layers)
        offs, total, bounds = {}, 0, []
        for ph in range(len(self.phases)):
            lo = total
            for n, p in named:
                if _bwd_phase(n, self.phases) == ph:
                    offs[n] = total
                    total += (p.numel() + 3) // 4 * 4
            bounds.append((lo, total))
        self.buckets = bounds
        q = getattr(self.m, "quantizer", None)
        self.n_stats = (q.K + q.K * q.D + 3) // 4 * 4 if q is not None else 0
        self.flat_p = torch.zeros(total, device=self.dev, dtype=torch.float32)
        # gradients, followed by the per-step EMA statistics [cnt(K) | sum(K*D)]: one RCCL all-reduce covers both
        self.flat_gx = torch.zeros(total + self.n_stats, device=self.dev, dtype=torch.float32)
        self.flat_g = self.flat_gx[:total]
        self.flat_m = torch.zeros(total, device=self.dev, dtype=torch.float32)
        self.flat_v = torch.zeros(total, device=self.dev, dtype=torch.float32)
        self.P, self.G = {}, {}
        with torch.no_grad():
            for n, p in named:
                o, k = offs[n], p.numel()
                view = self.flat_p[o:o + k].view(p.shape)
                view.copy_(p.detach().to(self.dev, torch.float32))
                p.data = view
                self.P[n] = view
                self.G[n] = self.flat_g[o:o + k].view(p.shape)
        self.n_flat = total
        self.offsets = offs

    def params_in_sync(self):
        for n, p in self.m.named_parameters():
            if p.data_ptr() != self.P[n].data_ptr():
                return False
        return True

    def attach_grads(self):
        """Expose the flat gradient as .grad views (drop-in optimizers / inspection)."""
        for n, p in self.m.named_parameters():
            p.grad = self.G[n]

    # ------------------------------------------------------------------ buffers / helpers
    def use_arena(self, key):
        """Make the arena of batch-shape `key` current (creating it if needed) and evict least-recently-used arenas --
        buffers AND the graphs that address them, together -- beyond max_arenas / max_arena_bytes."""
        a = self.arenas.get(key)
        if a is None:
            a = self.arenas[key] = _Arena(key)
        self.arenas.move_to_end(key)
        self.arena, self.buf = a, a.buf
        while len(self.arenas) > 1 and (len(self.arenas) > self.max_arenas or
                                        sum(x.nbytes for x in self.arenas.values()) > self.max_arena_bytes):
            old_key = next(iter(self.arenas))
            if old_key == key:
                break
            torch.cuda.synchronize(self.dev)        # nothing in flight may still address the evicted buffers
            self.arenas.pop(old_key).release()
        return a

    def bucket_len(self, L0):
        return bucket_length(L0, self.len_bucket, self.m.max_seq_len)

    def _stage_batch(self, x, mask):
        """Copy the batch into the bucketed input buffers of the current shape's arena (outside any graph): x [B, L0, 6] ->
        in.x [B, Lb, 6] zero padded, mask -> in.mask [B, Lb] with False on the padded tail.  x / mask may live on the host:
        a pinned batch (DataLoader pin_memory) then goes host -> arena in ONE asynchronous copy, queued behind the previous
        step's graph on the same stream (C2: 0.4 MB, ~10 us) -- there is no intermediate device tensor.  Returns (arena, x, mask); the
        mask stays None when the caller passed none and nothing was padded (the reference's mask=None variant)."""
        B, L0 = int(x.shape[0]), int(x.shape[1])
        # mask=None is the reference's un-masked variant (different normalisers): such a batch keeps its own length
        Lb = self.bucket_len(L0) if mask is not None else L0
        a = self.use_arena((B, Lb))
        xt = self.T("in.x", B, Lb, int(x.shape[2]))
        if Lb == L0:
            xt.copy_(x, non_blocking=True)
        else:
            xt[:, :L0].copy_(x, non_blocking=True)
            xt[:, L0:].zero_()
        ms = None
        if mask is not None or Lb != L0:
            ms = self.T("in.mask", B, Lb, dtype=torch.bool)
            if mask is not None:
                ms[:, :L0].copy_(mask, non_blocking=True)
            else:
                ms[:, :L0].fill_(True)
            if Lb != L0:
                ms[:, L0:].fill_(False)
        return a, xt, ms

    def T(self, name, *shape, dtype=torch.float32):
        """Named device buffer of the current arena.  (name, shape, dtype) always maps to the same storage, so a graph
        captured over it stays valid for the arena's lifetime; self.buf[name] is the latest tensor handed out as `name`."""
        a = self.arena
        k = (name, tuple(int(v) for v in shape), dtype)
        t = a.pool.get(k)
        if t is None:
            t = torch.empty(*k[1], device=self.dev, dtype=dtype)
            a.pool[k] = t
            a.nbytes += t.numel() * t.element_size()
        a.buf[name] = t
        return t

    def site(self, name):
        s = self._sites.get(name)
        if s is None:
            s = len(self._sites) + 1
            self._sites[name] = s
        return s

    def pdrop(self, p):
        return float(p) * self.drop_scale if self.train else 0.0

    # ---- primitive wrappers -------------------------------------------------------------------
    def ln_fwd(self, name, x, ldx, y, ldy, rows, tag):
        mean, rstd = self.T(tag + ".mean", rows), self.T(tag + ".rstd", rows)
        call("vqh_layernorm_fwd", x, ldx, self.P[name + ".weight"], self.P[name + ".bias"], y, ldy, mean, rstd, rows,
             self.H, LN_EPS)
        return mean, rstd

    def ln_bwd(self, name, dy, lddy, x, ldx, tag, dx, lddx, accumulate, rows, emit=None):
        """emit = (site, p, buffer tag): also write dx * keep-mask(site) -- the dropout backward of the block that consumes
        dx next -- and return that buffer (None when dropout is off)."""
        dxd, site, p = None, 0, 0.0
        if emit is not None and emit[1] > 0.0 and self.fold_dropout_bwd:
            site, p = emit[0], emit[1]
            dxd = self.TW(emit[2], rows * self.H)
        call("vqh_layernorm_bwd", dy, lddy, x, ldx, self.P[name + ".weight"], self.buf[tag + ".mean"],
             self.buf[tag + ".rstd"], dx, lddx, int(accumulate), self.G[name + ".weight"], self.G[name + ".bias"], 0.0,
             rows, self.H, dxd, self.rng if dxd is not None else None, site, p, self.ws, self.ws.numel())
        return dxd

    def lin_fwd(self, x, ldx, rows, W, b, out, ldo, mode=L.EPI_LINEAR, aux_in=None, aux_out=None, ldaux=0, site=0, p=0.0):
        N, K = W.shape
        # the workspace lets small-row GEMMs (the batch-shared first-layer projections: 64 rows) split K over the chip
        L.gemm(1, 1, rows, N, K, x, ldx, W, K, out, ldo, bias=b, mode=mode, aux_in=aux_in, aux_out=aux_out, ldaux=ldaux,
               rng=self.rng, site=site, p=p, ws=self.ws)

    def lin_dgrad(self, dy, lddy, rows, W, dx, lddx, mode=L.EPI_LINEAR, aux_in=None, ldaux=0, beta=0.0, p=0.0):
        N, K = W.shape          # dx[rows,K] = dy[rows,N] . W[N,K]
        L.gemm(1, 0, rows, K, N, dy, lddy, W, K, dx, lddx, mode=mode, aux_in=aux_in, ldaux=ldaux, beta=beta, p=p, ws=self.ws)

    def lin_wgrad(self, dy, lddy, x, ldx, rows, gW, gb):
        N, K = gW.shape         # gW[N,K] = dy[rows,N]^T . x[rows,K] ; gb[N] = column sums of dy (same launch)
        if self.group_wgrad:
            # deferred to flush_wgrads(): dy and x must stay untouched until then -- dy buffers come from dy_tag() (a fresh
            # one per block), saved activations are never written in backward (the FFN keeps d(pre-activation) apart)
            self._wq.append((dy, lddy, x, ldx, rows, gW, gb))
            return
        call("vqh_gemm_wgrad", rows, N, K, dy, lddy, x, ldx, gW, K, gb, 0.0, self.ws, self.ws.numel())

    def flush_wgrads(self):
        """Run the queued weight-gradient products of the layer just finished as one grouped launch."""
        if self._wq:
            from .parallel import dp_active
            if self.overlap_wgrad and not dp_active():
                main = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                    self._ws_side = torch.empty(WS_FLOATS, device=self.dev, dtype=torch.float32)
                self._side.wait_stream(main)              # operands and the gradients' earlier contents are ordered before it
                with torch.cuda.stream(self._side):
                    L.wgrad_group(self._wq, self._ws_side)
                ev = torch.cuda.Event()
                ev.record(self._side)
                for it in self._wq:                       # whoever rewrites an operand buffer waits for this launch (TW)
                    self._buf_ev[it[0].untyped_storage().data_ptr()] = ev
                self._side_busy = True
            else:
                L.wgrad_group(self._wq, self.ws)
            self._wq = []
        self._dy_slot = 0
        self._par ^= 1

    def join_wgrads(self):
        """The main stream waits for every weight-gradient launch of the second stream (before anything reads a gradient)."""
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_busy = False
        self._buf_ev.clear()

    def TW(self, name, *shape):
        """A scratch buffer that a DEFERRED weight-gradient product will read (its dY operand): two copies alternate from layer
        to layer, and a copy is handed out again only after the launch that read it (recorded in flush_wgrads)."""
        t = self.T(f"{name}@{self._par}" if self.overlap_wgrad else name, *shape)
        ev = self._buf_ev.pop(t.untyped_storage().data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return t

    def dy_tag(self):
        """Name of a fresh buffer for a block's output gradient (valid until the next flush_wgrads)."""
        t = f"tmp.dy{self._dy_slot}"
        self._dy_slot += 1
        return t

    def drop_bwd(self, dy, n, site, p, tag):
        """dy * keep-mask of a DROP_RESID site (identity when p == 0)."""
        if p <= 0.0:
            if not self.group_wgrad:
                return dy
            # the residual-stream gradient is updated in place later in the block, but a deferred weight-gradient product
            # still reads this value: give it its own copy
            out = self.TW(tag, n)
            call("vqh_copy2d", dy, n, out, n, 1, n)
            return out
        out = self.TW(tag, n)
        call("vqh_dropout_bwd", dy, out, n, self.rng, site, p)
        return out

    # ------------------------------------------------------------------ attention module
    def mha_fwd(self, pre, q_in, kv_in, rows_q, rows_kv, B, T, S, nh, kvalid, p_attn, self_attn, shared=False):
        """shared: the query-side input (and for self attention also the key/value input) is the same [T, H] block for
        every sample (first decoder / tokenizer layer); its projection runs on T rows instead of B*T."""
        H = self.H
        W, bias = self.P[pre + ".in_proj_weight"], self.P[pre + ".in_proj_bias"]
        pq = T if shared else rows_q
        if self_attn:
            qkv = self.T(pre + ".qkv", pq, 3 * H)
            self.lin_fwd(q_in, H, pq, W, bias, qkv, 3 * H)
            q, k, v, ldq, ldk = qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H
        else:
            qp = self.T(pre + ".q", pq, H)
            kvp = self.T(pre + ".kv", rows_kv, 2 * H)
            self.lin_fwd(q_in, H, pq, W[:H], bias[:H], qp, H)
            self.lin_fwd(kv_in, H, rows_kv, W[H:], bias[H:], kvp, 2 * H)
            q, k, v, ldq, ldk = qp, kvp, kvp[:, H:], H, 2 * H
        ao = self.T(pre + ".ao", rows_q, H)
        lse = self.T(pre + ".lse", B * nh * T)
        flag = (3 if self_attn else 1) if shared else 0
        call("vqh_attn_fwd", q, ldq, k, ldk, v, ldk, ao, H, lse, kvalid, B, nh, T, S, H // nh, flag, self.rng,
             self.site(pre + ".attn_drop"), p_attn)
        return ao

    def mha_bwd(self, pre, d_ao, q_in, kv_in, rows_q, rows_kv, B, T, S, nh, kvalid, p_attn, self_attn,
                d_q_in, d_kv_in, kv_beta, shared=False):
        """d_ao: grad of the head-concatenated attention output (before out_proj).
        Writes d_q_in (beta 0) and, for cross attention, d_kv_in (beta kv_beta).
        shared (see mha_fwd): the projection's weight and input gradients are linear in dQ(KV), so they are taken from
        its sum over the batch: d_q_in is then [T, H] = the batch-summed input gradient."""
        H = self.H
        flag = (3 if self_attn else 1) if shared else 0
        W = self.P[pre + ".in_proj_weight"]
        gW, gb = self.G[pre + ".in_proj_weight"], self.G[pre + ".in_proj_bias"]
        ao, lse = self.buf[pre + ".ao"], self.buf[pre + ".lse"]
        dsum = self.T("tmp.dsum", B * nh * T)
        site = self.site(pre + ".attn_drop")
        if self_attn:
            qkv = self.buf[pre + ".qkv"]
            dqkv = self.TW("tmp.dqkv", rows_q, 3 * H)
            call("vqh_attn_bwd", qkv, 3 * H, qkv[:, H:], 3 * H, qkv[:, 2 * H:], 3 * H, ao, H, lse, d_ao, H, dsum,
                 dqkv, 3 * H, dqkv[:, H:], 3 * H, dqkv[:, 2 * H:], 3 * H, kvalid, B, nh, T, S, H // nh, flag, self.rng, site, p_attn)
            if shared:
                dsh = self.TW("tmp.dqkv_sh", T, 3 * H)
                call("vqh_colsum", dqkv, T * 3 * H, B, T * 3 * H, dsh, 0.0, self.ws, self.ws.numel())
                dqkv, rows_q = dsh, T
            self.lin_wgrad(dqkv, 3 * H, q_in, H, rows_q, gW, gb)
            self.lin_dgrad(dqkv, 3 * H, rows_q, W, d_q_in, H)
        else:
            qp, kvp = self.buf[pre + ".q"], self.buf[pre + ".kv"]
            dq = self.TW("tmp.dq", rows_q, H)
            dkv = self.TW("tmp.dkv", rows_kv, 2 * H)
            call("vqh_attn_bwd", qp, H, kvp, 2 * H, kvp[:, H:], 2 * H, ao, H, lse, d_ao, H, dsum,
                 dq, H, dkv, 2 * H, dkv[:, H:], 2 * H, kvalid, B, nh, T, S, H // nh, flag, self.rng, site, p_attn)
            if shared:
                dsh = self.TW("tmp.dq_sh", T, H)
                call("vqh_colsum", dq, T * H, B, T * H, dsh, 0.0, self.ws, self.ws.numel())
                dq, rows_q = dsh, T
            self.lin_wgrad(dq, H, q_in, H, rows_q, gW[:H], gb[:H])
            self.lin_wgrad(dkv, 2 * H, kv_in, H, rows_kv, gW[H:], gb[H:])
            self.lin_dgrad(dq, H, rows_q, W[:H], d_q_in, H)
            self.lin_dgrad(dkv, 2 * H, rows_kv, W[H:], d_kv_in, H, beta=kv_beta)

    # ------------------------------------------------------------------ transformer blocks
    def attn_block_fwd(self, pre, attn, norm, x0, rows, B, T, nh, kvalid, mem=None, rows_kv=None, S=None, p=0.1,
                       shared=False):
        """x1 = x0 + dropout(out_proj(MHA(LN(x0) [, mem])));  shared: x0 repeats one [T, H] block per sample."""
        H = self.H
        nr = T if shared else rows
        h = self.T(f"{pre}.{norm}.y", nr, H)
        self.ln_fwd(f"{pre}.{norm}", x0, H, h, H, nr, f"{pre}.{norm}")
        self_attn = mem is None
        ao = self.mha_fwd(f"{pre}.{attn}", h, h if self_attn else mem, rows, rows if self_attn else rows_kv, B, T,
                          T if self_attn else S, nh, kvalid, self.pdrop(p), self_attn, shared=shared)
        x1 = self.T(f"{pre}.{attn}.out", rows, H)
        self.lin_fwd(ao, H, rows, self.P[f"{pre}.{attn}.out_proj.weight"], self.P[f"{pre}.{attn}.out_proj.bias"], x1, H,
                     mode=L.EPI_DROP_RESID, aux_in=x0, ldaux=H, site=self.site(f"{pre}.{attn}.drop"), p=self.pdrop(p))
        return x1

    def attn_block_bwd(self, pre, attn, norm, x0, dres, rows, B, T, nh, kvalid, mem=None, rows_kv=None, S=None,
                       d_mem=None, mem_beta=0.0, p=0.1, shared=False, dy_in=None, emit=None):
        """dres holds d x1 on entry and d x0 on exit (in place).
        shared: dres keeps only the residual branch; the LayerNorm branch's gradient, already summed over the batch
        (LayerNorm backward is linear in dy for fixed x), is returned as a [T, H] buffer for the caller to add to the
        gradient of the broadcast parameter."""
        H = self.H
        a = f"{pre}.{attn}"
        self_attn = mem is None
        h = self.buf[f"{pre}.{norm}.y"]
        # dy_in: dres * mask already written by the LayerNorm backward that produced dres (ln_bwd emit)
        dy = dy_in if dy_in is not None else self.drop_bwd(dres, rows * H, self.site(a + ".drop"), self.pdrop(p), self.dy_tag())
        self.lin_wgrad(dy, H, self.buf[a + ".ao"], H, rows, self.G[a + ".out_proj.weight"], self.G[a + ".out_proj.bias"])
        d_ao = self.T("tmp.dao", rows, H)
        self.lin_dgrad(dy, H, rows, self.P[a + ".out_proj.weight"], d_ao, H)
        dh = self.T("tmp.dh", rows, H)
        self.mha_bwd(a, d_ao, h, h if self_attn else mem, rows, rows if self_attn else rows_kv, B, T,
                     T if self_attn else S, nh, kvalid, self.pdrop(p), self_attn, dh, d_mem, mem_beta, shared=shared)
        if shared:
            dx_sh = self.T(f"{pre}.{norm}.dx_sh", T, H)
            self.ln_bwd(f"{pre}.{norm}", dh, H, x0, H, f"{pre}.{norm}", dx_sh, H, False, T)
            return dx_sh
        return self.ln_bwd(f"{pre}.{norm}", dh, H, x0, H, f"{pre}.{norm}", dres, H, True, rows, emit=emit)

    def ffn_block_fwd(self, pre, norm, lin1, lin2, x0, rows, act, p_inner, p_out):
        """x1 = x0 + dropout(lin2(dropout(act(lin1(LN(x0))))))   act: 'relu' (encoder/decoder) or 'gelu' (tokenizer)"""
        H = self.H
        h = self.T(f"{pre}.{norm}.y", rows, H)
        self.ln_fwd(f"{pre}.{norm}", x0, H, h, H, rows, f"{pre}.{norm}")
        W1, b1 = self.P[f"{pre}.{lin1}.weight"], self.P[f"{pre}.{lin1}.bias"]
        F = W1.shape[0]
        f1 = self.T(f"{pre}.{lin1}.y", rows, F)
        if act == "relu":
            self.lin_fwd(h, H, rows, W1, b1, f1, F, mode=L.EPI_RELU_DROP, site=self.site(f"{pre}.{lin1}.drop"),
                         p=self.pdrop(p_inner))
        else:
            pre_act = self.T(f"{pre}.{lin1}.pre", rows, F)
            self.lin_fwd(h, H, rows, W1, b1, f1, F, mode=L.EPI_GELU, aux_out=pre_act, ldaux=F)
        x1 = self.T(f"{pre}.{lin2}.out", rows, H)
        self.lin_fwd(f1, F, rows, self.P[f"{pre}.{lin2}.weight"], self.P[f"{pre}.{lin2}.bias"], x1, H,
                     mode=L.EPI_DROP_RESID, aux_in=x0, ldaux=H, site=self.site(f"{pre}.{lin2}.drop"), p=self.pdrop(p_out))
        return x1

    def ffn_block_bwd(self, pre, norm, lin1, lin2, x0, dres, rows, act, p_inner, p_out, dy_in=None, emit=None):
        H = self.H
        W1, W2 = self.P[f"{pre}.{lin1}.weight"], self.P[f"{pre}.{lin2}.weight"]
        F = W1.shape[0]
        f1 = self.buf[f"{pre}.{lin1}.y"]
        h = self.buf[f"{pre}.{norm}.y"]
        dy = dy_in if dy_in is not None else self.drop_bwd(dres, rows * H, self.site(f"{pre}.{lin2}.drop"), self.pdrop(p_out),
                                                           self.dy_tag())
        self.lin_wgrad(dy, H, f1, F, rows, self.G[f"{pre}.{lin2}.weight"], self.G[f"{pre}.{lin2}.bias"])
        if act == "relu":
            # d pre-activation: written over the saved post-activation (read-then-write per element), unless the weight
            # gradient of linear2 -- which needs that activation -- is deferred to the layer's grouped launch
            dpre = self.TW("tmp.dpre", rows, F) if self.group_wgrad else f1
            self.lin_dgrad(dy, H, rows, W2, dpre, F, mode=L.EPI_MUL_POSMASK, aux_in=f1, ldaux=F, p=self.pdrop(p_inner))
        else:
            dpre = self.buf[f"{pre}.{lin1}.pre"]
            self.lin_dgrad(dy, H, rows, W2, dpre, F, mode=L.EPI_MUL_GELUGRAD, aux_in=dpre, ldaux=F)
        self.lin_wgrad(dpre, F, h, H, rows, self.G[f"{pre}.{lin1}.weight"], self.G[f"{pre}.{lin1}.bias"])
        dh = self.T("tmp.dh", rows, H)
        self.lin_dgrad(dpre, F, rows, W1, dh, H)
        return self.ln_bwd(f"{pre}.{norm}", dh, H, x0, H, f"{pre}.{norm}", dres, H, True, rows, emit=emit)

    # ------------------------------------------------------------------ model sections
    def encode(self, x, mask):
        """models/vq_vae.py:639-660 -> h_fuse [B*L, H] (also keeps h_enc_geo / h_enc_ss buffers)."""
        B, Lq, _ = x.shape
        H, ML = self.H, B * Lq
        c = self.ctx
        c["x"], c["mask"], c["B"], c["L"] = x, mask, B, Lq
        pe = self.m.pos_enc[0]
        g = self.T("geo.x0", ML, H)
        call("vqh_embed_fwd", x, 6, 0, self.P["input_proj.weight"], self.P["input_proj.bias"], pe, g, ML, Lq, H,
             self.rng, self.site("inp_dropout"), self.pdrop(0.1))
        xs = [g]
        for i in range(self.m.num_layers):
            pre = f"encoder.layers.{i}"
            g = self.attn_block_fwd(pre, "self_attn", "norm1", g, ML, B, Lq, self.nh, mask)
            xs.append(g)
            g = self.ffn_block_fwd(pre, "norm2", "linear1", "linear2", g, ML, "relu", 0.1, 0.1)
            xs.append(g)
        c["geo_xs"] = xs
        h_geo = self.T("enc_ln.y", ML, H)
        self.ln_fwd("enc_ln", g, H, h_geo, H, ML, "enc_ln")
        cat = self.T("fuse.cat", ML, 2 * H)
        self.ln_fwd("ln_geo", h_geo, H, cat, 2 * H, ML, "ln_geo")
        s = self.T("ss.x0", ML, H)
        call("vqh_embed_fwd", x, 6, 3, self.P["ss_input_proj.weight"], self.P["ss_input_proj.bias"], pe, s, ML, Lq, H,
             self.rng, 0, 0.0)
        xs = [s]
        for i in range(SS_LAYERS):
            pre = f"ss_encoder.layers.{i}"
            s = self.attn_block_fwd(pre, "self_attn", "norm1", s, ML, B, Lq, self.nh, mask)
            xs.append(s)
            s = self.ffn_block_fwd(pre, "norm2", "linear1", "linear2", s, ML, "relu", 0.1, 0.1)
            xs.append(s)
        c["ss_xs"] = xs
        self.ln_fwd("ln_ss", s, H, cat[:, H:], 2 * H, ML, "ln_ss")
        f0, f0pre = self.T("fuse.f0", ML, H), self.T("fuse.f0pre", ML, H)
        self.lin_fwd(cat, 2 * H, ML, self.P["fuse_mlp.0.weight"], self.P["fuse_mlp.0.bias"], f0, H, mode=L.EPI_GELU,
                     aux_out=f0pre, ldaux=H)
        f2 = self.T("fuse.f2", ML, H)
        self.lin_fwd(f0, H, ML, self.P["fuse_mlp.2.weight"], self.P["fuse_mlp.2.bias"], f2, H)
        hf = self.T("fuse.out", ML, H)
        self.ln_fwd("fuse_mlp.3", f2, H, hf, H, ML, "fuse_mlp.3")
        return hf, h_geo, s

    def encode_bwd(self, d_hf):
        self.encode_bwd_ss(d_hf)
        for _ in self.encode_bwd_geo():
            pass

    def encode_bwd_ss(self, d_hf):
        """fuse_mlp and the secondary-structure encoder; leaves d(cat) for encode_bwd_geo."""
        c = self.ctx
        B, Lq, H = c["B"], c["L"], self.H
        ML, mask = B * Lq, c["mask"]
        df2 = self.T("tmp.df2", ML, H)
        self.ln_bwd("fuse_mlp.3", d_hf, H, self.buf["fuse.f2"], H, "fuse_mlp.3", df2, H, False, ML)
        self.lin_wgrad(df2, H, self.buf["fuse.f0"], H, ML, self.G["fuse_mlp.2.weight"], self.G["fuse_mlp.2.bias"])
        f0pre = self.buf["fuse.f0pre"]
        self.lin_dgrad(df2, H, ML, self.P["fuse_mlp.2.weight"], f0pre, H, mode=L.EPI_MUL_GELUGRAD, aux_in=f0pre, ldaux=H)
        cat = self.buf["fuse.cat"]
        self.lin_wgrad(f0pre, H, cat, 2 * H, ML, self.G["fuse_mlp.0.weight"], self.G["fuse_mlp.0.bias"])
        dcat = self.T("tmp.dcat", ML, 2 * H)
        self.lin_dgrad(f0pre, H, ML, self.P["fuse_mlp.0.weight"], dcat, 2 * H)
        # ---- secondary-structure branch
        xs = c["ss_xs"]
        dres = self.T("tmp.dres_seq", ML, H)
        dy0 = self.ln_bwd("ln_ss", dcat[:, H:], 2 * H, xs[-1], H, "ln_ss", dres, H, False, ML,
                          emit=(self.site(f"ss_encoder.layers.{SS_LAYERS - 1}.linear2.drop"), self.pdrop(0.1), "tmp.dy_carry_in")
                          if SS_LAYERS > 0 else None)
        for _ in self._encoder_stack_bwd("ss_encoder", SS_LAYERS, xs, dres, ML, B, Lq, mask, dy0=dy0):
            pass
        call("vqh_embed_bwd", dres, c["x"], 6, 3, self.G["ss_input_proj.weight"], self.G["ss_input_proj.bias"], 0.0, ML, H,
             self.rng, 0, 0.0, self.ws, self.ws.numel())

    def _encoder_stack_bwd(self, stack, n_layers, xs, dres, ML, B, Lq, mask, dy0=None):
        """Backward through n_layers pre-LN encoder layers.  Every block ends in a LayerNorm backward that accumulates into
        dres; it also emits dres * mask of the dropout site that opens the NEXT block (in backward order), so only the
        first block of the stack runs the stand-alone dropout-backward pass -- unless the caller's own LayerNorm backward
        already emitted it (dy0)."""
        dy = dy0
        for i in reversed(range(n_layers)):
            pre = f"{stack}.layers.{i}"
            dy = self.ffn_block_bwd(pre, "norm2", "linear1", "linear2", xs[2 * i + 1], dres, ML, "relu", 0.1, 0.1, dy_in=dy,
                                    emit=(self.site(f"{pre}.self_attn.drop"), self.pdrop(0.1), self.dy_tag()))
            nxt = None
            if i > 0:
                # consumed by the NEXT layer's first block, i.e. after this layer's flush: outside the per-layer rotation
                nxt = (self.site(f"{stack}.layers.{i - 1}.linear2.drop"), self.pdrop(0.1), f"tmp.dy_carry{i % 3}")
            dy = self.attn_block_bwd(pre, "self_attn", "norm1", xs[2 * i], dres, ML, B, Lq, self.nh, mask, dy_in=dy, emit=nxt)
            self.flush_wgrads()
            yield i                                   # layer i's gradients are final

    def encode_bwd_geo(self):
        """Generator: yields after every geometry-encoder layer (its gradients are final = one all-reduce bucket)."""
        c = self.ctx
        B, Lq, H = c["B"], c["L"], self.H
        ML, mask = B * Lq, c["mask"]
        dcat, dres = self.buf["tmp.dcat"], self.buf["tmp.dres_seq"]
        xs = c["geo_xs"]
        dhg = self.T("tmp.dh", ML, H)
        self.ln_bwd("ln_geo", dcat, 2 * H, self.buf["enc_ln.y"], H, "ln_geo", dhg, H, False, ML)
        nl = self.m.num_layers
        dy0 = self.ln_bwd("enc_ln", dhg, H, xs[-1], H, "enc_ln", dres, H, False, ML,
                          emit=(self.site(f"encoder.layers.{nl - 1}.linear2.drop"), self.pdrop(0.1), "tmp.dy_carry_in")
                          if nl > 0 else None)
        embed_done = False
        for i in self._encoder_stack_bwd("encoder", nl, xs, dres, ML, B, Lq, mask, dy0=dy0):
            if i == 0:                                # input_proj belongs to the last phase
                call("vqh_embed_bwd", dres, c["x"], 6, 0, self.G["input_proj.weight"], self.G["input_proj.bias"], 0.0, ML, H,
                     self.rng, self.site("inp_dropout"), self.pdrop(0.1), self.ws, self.ws.numel())
                embed_done = True
            yield i
        if not embed_done:                            # num_layers == 0
            call("vqh_embed_bwd", dres, c["x"], 6, 0, self.G["input_proj.weight"], self.G["input_proj.bias"], 0.0, ML, H,
                 self.rng, self.site("inp_dropout"), self.pdrop(0.1), self.ws, self.ws.numel())
            yield 0

    def tokenize(self, hf, mask, B, Lq):
        """LatentTokenizer + to_code (models/vq_vae.py:310-322, 736-743) -> z_e [B*N, D]"""
        H, N, D = self.H, self.N, self.D
        MN, ML = B * N, B * Lq
        c = self.ctx
        pd = self.m.tokenizer_dropout
        q = self.T("tok.q0", MN, H)
        call("vqh_bcast_rows", self.P["tokenizer.queries"], None, q, B, N * H)
        qs = [q]
        for i in range(self.m.tokenizer_layers):
            pre = f"tokenizer.layers.{i}"
            sh = self.share_layer0 and i == 0          # every sample starts from the same learned queries
            nq = N if sh else MN
            qn, kvn = self.T(pre + ".ln_q.y", nq, H), self.T(pre + ".ln_kv.y", ML, H)
            self.ln_fwd(pre + ".ln_q", q, H, qn, H, nq, pre + ".ln_q")
            self.ln_fwd(pre + ".ln_kv", hf, H, kvn, H, ML, pre + ".ln_kv")
            ao = self.mha_fwd(pre + ".attn", qn, kvn, MN, ML, B, N, Lq, self.tnh, mask, self.pdrop(pd), False, shared=sh)
            q1 = self.T(pre + ".attn.out", MN, H)
            self.lin_fwd(ao, H, MN, self.P[pre + ".attn.out_proj.weight"], self.P[pre + ".attn.out_proj.bias"], q1, H,
                         mode=L.EPI_DROP_RESID, aux_in=q, ldaux=H, site=self.site(pre + ".drop"), p=self.pdrop(pd))
            qs.append(q1)
            q = self.ffn_block_fwd(pre, "ln_o", "ffn.0", "ffn.2", q1, MN, "gelu", 0.0, pd)
            qs.append(q)
        c["tok_qs"] = qs
        z_e = self.T("tok.z_e", MN, D)
        sig = self.m.latent_sigmoid and ((not self.m.latent_sigmoid_ae_only) or (not self.m.use_vq))
        c["sigmoid"] = sig
        self.lin_fwd(q, H, MN, self.P["to_code.weight"], self.P["to_code.bias"], z_e, D,
                     mode=L.EPI_SIGMOID if sig else L.EPI_LINEAR)
        return z_e

    def tokenize_bwd(self, d_ze):
        """d_ze [B*N, D] (consumed) -> d_hf [B*L, H]"""
        c = self.ctx
        B, Lq, H, N, D = c["B"], c["L"], self.H, self.N, self.D
        MN, ML, mask = B * N, B * Lq, c["mask"]
        pd = self.m.tokenizer_dropout
        qs = c["tok_qs"]
        if c["sigmoid"]:
            call("vqh_sigmoid_bwd", d_ze, self.buf["tok.z_e"], d_ze, MN * D)
        self.lin_wgrad(d_ze, D, qs[-1], H, MN, self.G["to_code.weight"], self.G["to_code.bias"])
        dres = self.T("tmp.dres_tok", MN, H)
        self.lin_dgrad(d_ze, D, MN, self.P["to_code.weight"], dres, H)
        d_hf = self.T("tmp.d_hf", ML, H)
        nl = self.m.tokenizer_layers
        for i in reversed(range(nl)):
            pre = f"tokenizer.layers.{i}"
            a = pre + ".attn"
            dy = self.ffn_block_bwd(pre, "ln_o", "ffn.0", "ffn.2", qs[2 * i + 1], dres, MN, "gelu", 0.0, pd,
                                    emit=(self.site(pre + ".drop"), self.pdrop(pd), self.dy_tag()))
            if dy is None:
                dy = self.drop_bwd(dres, MN * H, self.site(pre + ".drop"), self.pdrop(pd), self.dy_tag())
            self.lin_wgrad(dy, H, self.buf[a + ".ao"], H, MN, self.G[a + ".out_proj.weight"], self.G[a + ".out_proj.bias"])
            d_ao = self.T("tmp.dao_tok", MN, H)
            self.lin_dgrad(dy, H, MN, self.P[a + ".out_proj.weight"], d_ao, H)
            dqn, dkvn = self.T("tmp.dqn", MN, H), self.T("tmp.dkvn", ML, H)
            sh = self.share_layer0 and i == 0
            self.mha_bwd(a, d_ao, self.buf[pre + ".ln_q.y"], self.buf[pre + ".ln_kv.y"], MN, ML, B, N, Lq, self.tnh, mask,
                         self.pdrop(pd), False, dqn, dkvn, 0.0, shared=sh)
            if sh:      # dqn[:N] is the batch-summed gradient; the LayerNorm branch goes straight to the parameter grad
                dq_sh = self.T("tmp.dq0_sh", N, H)
                self.ln_bwd(pre + ".ln_q", dqn, H, qs[0], H, pre + ".ln_q", dq_sh, H, False, N)
            else:
                self.ln_bwd(pre + ".ln_q", dqn, H, qs[2 * i], H, pre + ".ln_q", dres, H, True, MN)
            self.ln_bwd(pre + ".ln_kv", dkvn, H, self.buf["fuse.out"], H, pre + ".ln_kv", d_hf, H, i != nl - 1, ML)
            self.flush_wgrads()
        self.flush_wgrads()
        gq = self.G["tokenizer.queries"]
        call("vqh_colsum", dres, N * H, B, N * H, gq, 0.0, self.ws, self.ws.numel())
        if self.share_layer0 and nl > 0:
            call("vqh_add", gq, dq_sh, gq, N * H)
        return d_hf

    def _soft_vq_params(self):
        """(tau, alpha) of the soft-VQ branch (:835-836, :849) or None when the hard path is taken."""
        m = self.m
        if not (m.use_vq and m.soft_vq_use and self.train and m.num_quantizers == 1):
            return None
        ws = m.soft_vq_tau_warm_steps
        if ws <= 0:
            tau = m.soft_vq_tau_end
        else:
            t = min(1.0, max(0.0, m.training_steps) / float(ws))
            tau = (1.0 - t) * m.soft_vq_tau_start + t * m.soft_vq_tau_end
        aw = m.soft_vq_alpha_warm_steps
        alpha = 1.0 if aw <= 0 else min(1.0, float(m.training_steps) / float(aw))
        return float(tau), float(alpha)

    def _soft_vq_key(self):
        p = self._soft_vq_params() if self.m.use_vq else None
        return p

    def quantize(self, z_e, B, do_ema_update, row_valid=None):
        """quantize_gen driven to completion, with the per-level statistics all-reduce done inline (eager callers)."""
        gen = self.quantize_gen(z_e, B, do_ema_update, row_valid)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value
            torch.distributed.all_reduce(self.flat_gx[self.n_flat:])      # RCCL sum of the EMA statistics over ranks

    def quantize_gen(self, z_e, B, do_ema_update, row_valid=None):
        """Generator form: yields where the EMA statistics [cnt | sum] of a level must be summed over the ranks BEFORE that
        level's table refresh (world > 1 and not deferred: residual VQ :251-258, or the usage-entropy regulariser) -- the
        caller all-reduces self.flat_gx[self.n_flat:], between two graph segments when the step is captured.

        VectorQuantizerEMA.forward (models/vq_vae.py:170-283). z_e [B*N, D] -> z_st, z_q, idx, stats
        row_valid [B*N] bool (the optional `mask` of :175): only valid positions feed the EMA statistics (:192-197,
        :251-256) and, single level only, the usage histogram (:202-205); VQVAE.forward never passes one (:869)."""
        q = self.m.quantizer
        D, R = self.D, z_e.shape[0]
        Q, Kp, K = q.num_quantizers, q.K_per, q.K
        emb = q.embedding
        idx = self.T("vq.idx", Q * R, dtype=torch.int64)
        zq_lv = self.T("vq.zq_levels", Q, R, D)
        cnt = self.flat_gx[self.n_flat:self.n_flat + K]
        ssum = self.flat_gx[self.n_flat + K:self.n_flat + K + K * D].view(K, D)
        usage = self.T("vq.usage", K)
        soft_on = self._soft_vq_params() is not None
        defer = bool((self.defer_ema and Q == 1) or soft_on)      # soft-VQ probabilities use the pre-refresh table
        upd = bool(self.train and do_ema_update)
        if row_valid is not None:
            row_valid = row_valid.reshape(-1).contiguous()
            if not bool(row_valid.any()):          # no valid position: the reference skips the update (:196, :253)
                upd = False
        from .parallel import dp_active
        multi = dp_active()
        rows = z_e
        res = [self.T("vq.res0", R, D), self.T("vq.res1", R, D)]
        # the plane-tensor score form wants room for the split Z / codebook images; the workspace can only move while no
        # captured graph holds its address (eager steps come first)
        want = L.vq_nearest_workspace(R, Kp, D)
        if (want > self.ws.numel() and not torch.cuda.is_current_stream_capturing()
                and not any(a.graphs for a in self.arenas.values())):
            self.ws = torch.empty(want, device=self.dev, dtype=torch.float32)
        for lv in range(Q):
            lo = lv * Kp
            tab = emb[lo:lo + Kp]
            ids = idx[lv * R:(lv + 1) * R]
            call("vqh_vq_nearest", rows, D, tab, D, ids, lo, R, Kp, D, 3e-5, self.ws, self.ws.numel())
            nxt = res[lv & 1] if lv + 1 < Q else None
            # gather uses the table BEFORE this level's EMA refresh (:189 precedes :191-197 / :248 precedes :251-258)
            call("vqh_vq_gather", tab, D, ids, lo, rows, D, zq_lv[lv], nxt, R, D)
            # per-code statistics of THIS level (other code ranges stay zero: the reference refreshes the whole table)
            call("vqh_memset", cnt, 0, K * 4)
            call("vqh_memset", ssum, 0, K * D * 4)
            ids_stat = ids
            if row_valid is not None:
                if Q > 1:       # the residual branch histograms every position (:264): count before masking
                    call("vqh_vq_segment_sum", rows, D, ids, R, D, lo, Kp, cnt, ssum, self.ws, self.ws.numel())
                    call("vqh_copy2d", cnt[lo:], Kp, usage[lo:], Kp, 1, Kp)
                    call("vqh_memset", cnt, 0, K * 4)
                    call("vqh_memset", ssum, 0, K * D * 4)
                ids_stat = self.T("vq.ids_valid", R, dtype=torch.int64)
                call("vqh_vq_mask_ids", ids, row_valid, ids_stat, R)
            call("vqh_vq_segment_sum", rows, D, ids_stat, R, D, lo, Kp, cnt, ssum, self.ws, self.ws.numel())
            if row_valid is None or Q == 1:
                call("vqh_copy2d", cnt[lo:], Kp, usage[lo:], Kp, 1, Kp)
            if upd and defer:
                # single level: this step's z_q already used the old table (:189 precedes :191-197), so the refresh
                # may run after backward, behind the combined gradient + statistics all-reduce
                self._pending_ema = float(q.decay)
            elif upd:
                if multi:                                   # RCCL sum of the EMA statistics over ranks
                    yield "stats"
                self.apply_ema(float(q.decay))
            rows = nxt
        z_q, z_st = self.T("vq.z_q", R, D), self.T("vq.z_st", R, D)
        call("vqh_vq_finish", zq_lv, Q, z_e, D, z_q, z_st, R, D)
        soft = self._soft_vq_params()
        if soft is not None:
            # soft-VQ (:828-861): decoder input = z_e + ((1-a) softmax(-d2/tau).E + a z_hard - z_e), value only
            tau, alpha = soft
            tau = max(1e-8, tau)
            S = self.T("vq.soft", R, K)
            L.gemm(1, 1, R, K, D, z_e, D, emb, D, S, K)
            cb = self.T("vq.soft_bias", K)
            call("vqh_row_sqnorm", emb, D, K, D, cb, -1.0 / tau)
            call("vqh_softmax_rows", S, K, cb, 2.0 / tau, R, K)         # -(|z|^2 - 2 z.e + |e|^2)/tau up to a row constant
            zs = self.T("vq.z_soft", R, D)
            L.gemm(1, 0, R, D, K, S, K, emb, D, zs, D)
            call("vqh_vq_mix", z_e, zs, z_q, alpha, z_st, R * D)
            call("vqh_vq_usage_stats", usage, K, float(Q * R), None, None, self.vq_stats)   # _compute_stats: no epoch sums
            if not self.defer_ema:
                self.finish_ema()
            return z_st, z_q, idx, self.vq_stats
        call("vqh_vq_usage_stats", usage, K, float(Q * R), q._ep_usage, q._ep_cnt, self.vq_stats)
        return z_st, z_q, idx, self.vq_stats

    def maybe_reinit_dead_codes(self):
        """Trigger of VQVAE.forward (models/vq_vae.py:874-891) + VectorQuantizerEMA._maybe_reinit_dead_codes (:91-107).
        Runs eagerly after the step (the step's z_q was gathered before, so only later steps see the new codes).
        One host sync every 500 steps, like the reference's `.item()`.  With several ranks, rank 0's re-seeded
        buffers are broadcast (the reference's DDP broadcasts rank 0's buffers every forward)."""
        m, q = self.m, self.m.quantizer
        if q is None or not self.train:
            return False
        step = m.training_steps
        if not (step % 500 == 0 and step >= max(m.ema_update_freeze_steps, 800)):
            return False
        if step < m.ema_update_freeze_steps or not q.reinit_dead_codes or q.reinit_prob <= 0.0:
            return False
        usage = self.buf["vq.usage"]
        if self.world() > 1:
            usage = usage.clone()
            torch.distributed.all_reduce(usage)
        n_dead = int((usage <= float(q.dead_usage_threshold)).sum().item())
        R = self.buf["tok.z_e"].shape[0]
        if n_dead <= 0 or R == 0:
            return False
        if float(torch.rand(())) > q.reinit_prob:
            return False
        pick = torch.randint(0, R, (q.K,), device=self.dev)
        call("vqh_vq_reinit", usage, float(q.dead_usage_threshold), pick, self.buf["tok.z_e"], self.D, q.embedding,
             q.ema_embedding, q.ema_cluster_size, q.K, self.D)
        if self.world() > 1:
            for t in (q.embedding, q.ema_embedding, q.ema_cluster_size):
                torch.distributed.broadcast(t, 0)
        return True

    def apply_ema(self, decay):
        q = self.m.quantizer
        K, D = q.K, q.D
        cnt = self.flat_gx[self.n_flat:self.n_flat + K]
        ssum = self.flat_gx[self.n_flat + K:self.n_flat + K + K * D]
        call("vqh_vq_ema_apply", cnt, ssum, q.ema_cluster_size, q.ema_embedding, q.embedding, K, D,
             _f32(decay), _f32(1.0 - decay), VQ_EPS)

    def finish_ema(self):
        if self._pending_ema is not None:
            self.apply_ema(self._pending_ema)
            self._pending_ema = None

    def decode(self, z, mask, B, Lq):
        """models/vq_vae.py:745-765. z [B*N, D] -> recons [B*L, 6]"""
        H, N, D = self.H, self.N, self.D
        MN, ML = z.shape[0], B * Lq
        Nmem = MN // B
        c = self.ctx
        c["dec_B"], c["dec_L"], c["dec_N"], c["dec_z"], c["dec_mask"] = B, Lq, Nmem, z, mask
        memf = self.T("dec.memf", MN, H)
        self.lin_fwd(z, D, MN, self.P["from_code.weight"], self.P["from_code.bias"], memf, H)
        mem = self.T("dec.mem", MN, H)
        self.ln_fwd("mem_ln", memf, H, mem, H, MN, "mem_ln")
        t = self.T("dec.x0", ML, H)
        call("vqh_bcast_rows", self.P["query_embed.weight"], self.m.pos_enc[0], t, B, Lq * H)
        xs = [t]
        for i in range(self.m.num_layers):
            pre = f"decoder.layers.{i}"
            t = self.attn_block_fwd(pre, "self_attn", "norm1", t, ML, B, Lq, self.nh, mask,
                                    shared=self.share_layer0 and i == 0)
            xs.append(t)
            t = self.attn_block_fwd(pre, "multihead_attn", "norm2", t, ML, B, Lq, self.nh, None, mem=mem, rows_kv=MN, S=Nmem)
            xs.append(t)
            t = self.ffn_block_fwd(pre, "norm3", "linear1", "linear2", t, ML, "relu", 0.1, 0.1)
            xs.append(t)
        c["dec_xs"] = xs
        rec = self.T("dec.recons", ML, 6)
        self.lin_fwd(t, H, ML, self.P["head_xyz.weight"], self.P["head_xyz.bias"], rec, 6)
        self.lin_fwd(t, H, ML, self.P["head_ss.weight"], self.P["head_ss.bias"], rec[:, 3:], 6)
        return rec

    def decode_bwd(self, d_rec, d_z, z_beta):
        for _ in self.decode_bwd_gen(d_rec, d_z, z_beta):
            pass

    def decode_bwd_gen(self, d_rec, d_z, z_beta):
        """d_rec [B*L, 6] -> d_z [B*N, D] written as z_beta*d_z + grad (straight-through adds onto the commitment grad).
        Generator: yields after every decoder layer (phase boundary); the last yield follows the memory / query tail."""
        c = self.ctx
        B, Lq, Nmem, H, D = c["dec_B"], c["dec_L"], c["dec_N"], self.H, self.D
        ML, MN, mask = B * Lq, B * c["dec_N"], c["dec_mask"]
        xs = c["dec_xs"]
        t = xs[-1]
        self.lin_wgrad(d_rec, 6, t, H, ML, self.G["head_xyz.weight"], self.G["head_xyz.bias"])
        self.lin_wgrad(d_rec[:, 3:], 6, t, H, ML, self.G["head_ss.weight"], self.G["head_ss.bias"])
        dres = self.T("tmp.dres_seq", ML, H)
        self.lin_dgrad(d_rec, 6, ML, self.P["head_xyz.weight"], dres, H)
        self.lin_dgrad(d_rec[:, 3:], 6, ML, self.P["head_ss.weight"], dres, H, beta=1.0)
        mem = self.buf["dec.mem"]
        d_mem = self.T("tmp.d_mem", MN, H)
        nl = self.m.num_layers
        dy, dx_sh, pd = None, None, self.pdrop(0.1)
        for i in reversed(range(nl)):
            pre = f"decoder.layers.{i}"
            dy = self.ffn_block_bwd(pre, "norm3", "linear1", "linear2", xs[3 * i + 2], dres, ML, "relu", 0.1, 0.1, dy_in=dy,
                                    emit=(self.site(f"{pre}.multihead_attn.drop"), pd, self.dy_tag()))
            dy = self.attn_block_bwd(pre, "multihead_attn", "norm2", xs[3 * i + 1], dres, ML, B, Lq, self.nh, None, mem=mem,
                                     rows_kv=MN, S=Nmem, d_mem=d_mem, mem_beta=0.0 if i == nl - 1 else 1.0, dy_in=dy,
                                     emit=(self.site(f"{pre}.self_attn.drop"), pd, self.dy_tag()))
            sh0 = self.share_layer0 and i == 0
            nxt = (self.site(f"decoder.layers.{i - 1}.linear2.drop"), pd, f"tmp.dy_carry{i % 3}") if i > 0 else None
            out = self.attn_block_bwd(pre, "self_attn", "norm1", xs[3 * i], dres, ML, B, Lq, self.nh, mask, shared=sh0,
                                      dy_in=dy, emit=None if sh0 else nxt)
            if sh0:
                dx_sh, dy = out, None
            else:
                dy = out
            if i > 0:
                self.flush_wgrads()
                yield i
        # tgt = query_embed[:L] + pos_enc[:L] broadcast over the batch
        gq = self.G["query_embed.weight"]
        call("vqh_memset", gq, 0, gq.numel() * 4)
        call("vqh_colsum", dres, Lq * H, B, Lq * H, gq, 0.0, self.ws, self.ws.numel())
        if dx_sh is not None:
            call("vqh_add", gq, dx_sh, gq, Lq * H)
        dmemf = self.TW("tmp.dmemf", MN, H)
        self.ln_bwd("mem_ln", d_mem, H, self.buf["dec.memf"], H, "mem_ln", dmemf, H, False, MN)
        self.lin_wgrad(dmemf, H, c["dec_z"], D, MN, self.G["from_code.weight"], self.G["from_code.bias"])
        self.lin_dgrad(dmemf, H, MN, self.P["from_code.weight"], d_z, D, beta=z_beta)
        self.flush_wgrads()
        yield 0

    # ------------------------------------------------------------------ whole step
    def forward(self, x, mask, train=True):
        """VQVAE.forward (models/vq_vae.py:767-901) without aug/noise/soft-VQ (see DESIGN.md scope)."""
        self.train = bool(train)
        self.defer_ema = False
        self.use_arena((int(x.shape[0]), int(x.shape[1])))
        x_in = self.augment_input(x)
        upd = self._host_prologue()
        out = self._forward_core(x_in, mask, upd)
        self.finish_ema()
        if upd:
            self.maybe_reinit_dead_codes()
        return out

    def augment_input(self, x):
        """Input-only rigid augmentation and coordinate noise (:775-792). Random draws come from torch's generators
        like the reference (host gate torch.rand(()), device torch.rand / randn), the transform runs in vqh_augment."""
        m = self.m
        if not self.train:
            return x
        rot = m.rigid_aug_prob > 0.0 and float(torch.rand(())) < m.rigid_aug_prob
        std = 0.0
        if m.max_noise_std > 0.0:
            f = min(1.0, m.training_steps / float(m.noise_warmup_steps)) if m.noise_warmup_steps > 0 else 1.0
            std = m.max_noise_std * f
        if not rot and std <= 0.0:
            return x
        B, Lq = x.shape[0], x.shape[1]
        u = t = noise = None
        if rot:
            u = torch.stack([torch.rand(B, device=self.dev) for _ in range(3)], dim=1).contiguous()
            t = (torch.randn(B, 1, 3, device=self.dev) * 0.02).reshape(B, 3).contiguous()
        if std > 0.0:
            noise = (torch.randn(B, Lq, 3, device=self.dev) * std).contiguous()
        out = self.T("in.x_aug", B, Lq, 6)
        call("vqh_augment", x, u, t, noise, out, B, Lq)
        return out

    def _host_prologue(self):
        """Host-side state of one forward: EMA decay schedule (:795-802), step counter (:805-806), freeze (:825)."""
        m = self.m
        if m.use_vq:
            ws = m.ema_decay_warm_steps
            if m._ema_decay_override is not None:
                m.quantizer.decay = float(m._ema_decay_override)
            elif ws <= 0:
                m.quantizer.decay = float(m.ema_decay_end)
            else:
                t = min(1.0, max(0.0, m.training_steps) / float(ws))
                m.quantizer.decay = float((1.0 - t) * m.ema_decay_start + t * m.ema_decay_end)
        if self.train:
            m.training_steps += 1
        return bool(self.train and m.use_vq and (m.training_steps >= m.ema_update_freeze_steps))

    def _forward_core(self, x, mask, upd):
        gen = self._forward_core_gen(x, mask, upd)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value
            torch.distributed.all_reduce(self.flat_gx[self.n_flat:])

    def _forward_core_gen(self, x, mask, upd):
        m = self.m
        self.ctx = {}
        self.fwd_id += 1
        B, Lq, _ = x.shape
        hf, _, _ = self.encode(x, mask)
        z_e = self.tokenize(hf, mask, B, Lq)
        c = self.ctx
        if not m.use_vq:
            z_dec, z_q, idx, stats = z_e, z_e, None, None
        else:
            z_dec, z_q, idx, stats = yield from self.quantize_gen(z_e, B, upd)
        rec = self.decode(z_dec, mask, B, Lq)
        c["z_e"], c["z_q"], c["idx"], c["rec"], c["stats"] = z_e, z_q, idx, rec, stats
        return rec, z_e, z_q, idx, stats

    def loss(self, rec, target, mask, z_e, z_q, stats, weights, L_logical=None, L_dev=None):
        """VQVAE.loss_function (models/vq_vae.py:1097-1388): metrics (device vector) + d_rec, d_ze.
        target is [B, Ls, 6] in memory; the loss is evaluated on the reference's view [B, L] with L = the batch's own
        padded length (pad_collate's L_max: the window / long-range-pair enumerations of :996-1095 depend on it), given as
        L_logical (host int) or L_dev (one device float, so that a captured graph serves every L_max of its bucket)."""
        m = self.m
        B, Ls = target.shape[0], target.shape[1]
        Lq = int(L_logical) if L_logical is not None else Ls
        stats6 = None
        if m._data_std is not None:                                   # set_data_stats(): to_real() of :1218-1227
            mean = m._data_mean if m._data_mean is not None else torch.zeros(3)
            stats6 = [float(v) for v in m._data_std.reshape(-1).tolist()] + [float(v) for v in mean.reshape(-1).tolist()]
        g = lambda k, d: float(weights.get(k, d))
        w = [g("rmsd_weight", 1.0), g("ss_weight", 1.0), g("bond_length_weight", 0.0), g("bond_angle_weight", 0.0),
             g("dir_weight", 0.0), g("dih_weight", 0.0), g("xyz_tv_lambda", 0.0), g("pdm_weight", 0.0),
             g("win_kabsch_weight", 0.0), g("kappa_weight", 0.0), g("tau_weight", 0.0), g("lr_pdm_weight", 0.0),
             float(m.xyz_align_alpha), float(m.ss_tv_lambda), float(m.label_smoothing or 0.0),
             float(m.quantizer.beta) if m.use_vq else 0.0]
        ip = [int(weights.get("pdm_window", 8)), int(weights.get("win_kabsch_size", 16)),
              int(weights.get("win_kabsch_stride", 8)), int(weights.get("lr_min_sep", 24)),
              int(weights.get("lr_stride", 8)), int(weights.get("lr_max_offsets", 8))]
        import ctypes as C
        wa = (C.c_float * 16)(*w)
        ia = (C.c_int * 6)(*ip)
        sa = (C.c_float * 6)(*stats6) if stats6 is not None else None
        d_rec = self.T("loss.d_rec", B * Ls, 6)
        use_vq = bool(m.use_vq)
        Ntok = z_e.shape[0] // B
        d_ze = self.T("loss.d_ze", z_e.shape[0], self.D)
        call("vqh_loss_fwd_bwd", rec, target, mask, 1 if mask is not None else 0, z_e if use_vq else None,
             z_q if use_vq else None, stats if use_vq else None, B, Lq, Ls, L_dev, Ntok, self.D, int(use_vq),
             C.cast(wa, C.c_void_p).value, C.cast(ia, C.c_void_p).value,
             C.cast(sa, C.c_void_p).value if sa is not None else None, d_rec, d_ze if use_vq else None, self.metrics,
             self.ws, self.ws.numel())
        lam = float(m.usage_entropy_lambda)
        if lam > 0.0 and use_vq and z_e.shape[0] > 0:
            # usage-entropy regulariser (:1299-1309): softmax(z_e . E^T) -> mean code probability -> -lambda * entropy,
            # with its gradient added onto d_ze; uses the table as it stands after this step's EMA refresh
            q = m.quantizer
            R, K = z_e.shape[0], q.K
            logits = self.T("ue.logits", R, K)
            L.gemm(1, 1, R, K, self.D, z_e, self.D, q.embedding, self.D, logits, K)
            call("vqh_softmax_rows", logits, K, None, 1.0, R, K)
            pc, gk = self.T("ue.pc", K), self.T("ue.g", K)
            call("vqh_colsum", logits, K, R, K, pc, 0.0, self.ws, self.ws.numel())
            call("vqh_usage_entropy_finish", pc, K, R, lam, gk, self.metrics, 0, METRIC_KEYS.index("Usage_Reg"))
            call("vqh_softmax_bwd_colgrad", logits, K, gk, R, K)
            L.gemm(1, 0, R, self.D, K, logits, K, q.embedding, self.D, d_ze, self.D, beta=1.0)
        if self.ctx is None:
            self.ctx = {}
        self.ctx["d_rec"], self.ctx["d_ze"], self.ctx["weights"] = d_rec, d_ze, dict(weights)
        return self.metrics

    def backward(self):
        """Fill the flat gradient buffer from the d_rec / d_ze left by loss()."""
        for _ in self.backward_gen():
            pass

    def backward_gen(self):
        """Backward as a generator over the phases of bwd_phases(): after the k-th yield, bucket self.buckets[k] of the flat
        gradient is final (its all-reduce may start)."""
        c = self.ctx
        self._par = 0                                   # the same buffer names in every backward (graphs replay their own)
        for _ in self.decode_bwd_gen(c["d_rec"], c["d_ze"], 1.0 if self.m.use_vq else 0.0):
            yield
        c["d_hf"] = self.tokenize_bwd(c["d_ze"])
        self.flush_wgrads()
        yield
        self.encode_bwd_ss(c["d_hf"])
        self.flush_wgrads()
        yield
        for _ in self.encode_bwd_geo():
            yield
        self.join_wgrads()

    def set_hyper(self, lr, weight_decay, max_norm, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, L_logical=0):
        """Host -> device scalars of the next optimizer step (copied on the current stream, outside any graph)."""
        self.opt_step += 1
        t = self.opt_step
        # pinned staging ring + asynchronous copy: a pageable-memory copy would make the host wait for the previous step's
        # graph every step; the ring's events bound the host's run-ahead to HYPER_SLOTS steps instead
        if self._hyper_host is None:
            self._hyper_host = [torch.zeros(10, dtype=torch.float32).pin_memory() for _ in range(HYPER_SLOTS)]
            self._hyper_ev = [None] * HYPER_SLOTS
        slot = t % HYPER_SLOTS
        if self._hyper_ev[slot] is not None:
            self._hyper_ev[slot].synchronize()
        hb = self._hyper_host[slot]
        vals = [lr, betas[0], betas[1], eps, weight_decay, max_norm if max_norm else 0.0,
                1.0 - betas[0] ** t, 1.0 - betas[1] ** t, grad_scale, float(L_logical)]
        for i, v in enumerate(vals):
            hb[i] = float(v)
        self.hyper.copy_(hb, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_ev[slot] = ev

    def optimizer_step(self):
        """clip_grad_norm_(max_norm) + AdamW over the flat buffers (hyper-parameters from self.hyper)."""
        call("vqh_grad_norm", self.flat_g, self.n_flat, self.hyper, self.norm, self.norm_ws)
        call("vqh_adamw_step", self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.n_flat, self.hyper, self.norm)
        self._grads_pending = False                 # consumed: the next backward() overwrites instead of accumulating

    def world(self):
        from .parallel import world_size
        return world_size()

    def allreduce_grads(self):
        """Blocking form (all buckets at once): sum over ranks of [gradients | EMA statistics]; the 1/world gradient
        average is folded into the clip coefficient (hyper[8]), the statistics are wanted as sums."""
        from .parallel import allreduce_flat
        allreduce_flat(self.flat_gx, include_stats=self._pending_ema is not None, n_grad=self.n_flat)

    def allreduce_bucket(self, ph, works):
        """Start the sum all-reduce of bucket ph (the last one carries the EMA statistics behind it) without waiting:
        RCCL runs it on its own stream behind everything queued so far, next to the following backward phase."""
        from .parallel import allreduce_async
        lo, hi = self.buckets[ph]
        if ph == len(self.buckets) - 1 and self._pending_ema is not None:
            hi = self.n_flat + self.n_stats
        w = allreduce_async(self.flat_gx[lo:hi])
        if w is not None:
            works.append(w)

    # ------------------------------------------------------------------ fused training step
    def _step_gen(self, x_in, x_tgt, mask, weights, upd):
        """The whole training step as a generator.  Every yield is a point where a collective has to run when world > 1:
            ("stats",)      sum the EMA statistics of one quantizer level before its table refresh (residual VQ /
                            usage-entropy regulariser only; a single level defers its refresh behind the last bucket)
            ("bucket", k)   gradients of backward phase k are final: start the all-reduce of bucket k
            ("wait",)       all buckets must have arrived: the optimizer follows
        A single process ignores the yields (one graph holds everything); with several ranks the stretches between two
        yields become separate hipGraph segments and the collectives run between their replays (StepEngine._replay)."""
        self.advance_rng()
        fwd = self._forward_core_gen(x_in, mask, upd)
        while True:
            try:
                next(fwd)
            except StopIteration as stop:
                rec, z_e, z_q, idx, stats = stop.value
                break
            yield ("stats",)
        self.loss(rec, x_tgt, mask, z_e, z_q, stats, weights, L_dev=self.hyper[9:])   # L_max of THIS batch: set_hyper
        ph = 0
        for _ in self.backward_gen():
            yield ("bucket", ph)
            ph += 1
        yield ("wait",)
        self._step_part_b()

    def _act(self, action, works):
        """Run one collective point of _step_gen (world > 1)."""
        if action[0] == "stats":
            torch.distributed.all_reduce(self.flat_gx[self.n_flat:])
        elif action[0] == "bucket":
            self.allreduce_bucket(action[1], works)
        elif action[0] == "wait":
            for w in works:
                w.wait()
            del works[:]

    def _step_eager(self, x_in, x_tgt, mask, weights, upd, dp):
        works = []
        for action in self._step_gen(x_in, x_tgt, mask, weights, upd):
            if dp:
                self._act(action, works)

    def _step_part_b(self):
        self.finish_ema()
        self.optimizer_step()
        call("vqh_add", self.metrics_acc, self.metrics, self.metrics_acc, self.metrics.numel())

    def train_step(self, x, mask, weights, lr, weight_decay, clip, use_graph=True):
        out = self._train_step(x, mask, weights, lr, weight_decay, clip, use_graph)
        if self.m.use_vq and self.m.training_steps % 500 == 0:
            self.maybe_reinit_dead_codes()
        return out

    def _replay(self, segs, dp, upd, decay):
        """segs = [(graph, action after it)]: one graph for a single process, one per stretch between collectives else."""
        if not dp:
            segs[0][0].replay()
            return
        # host-side state the captured forward set while being recorded: a deferred single-level EMA refresh rides
        # behind the last gradient bucket (allreduce_bucket appends the statistics when _pending_ema is set)
        self._pending_ema = decay if (upd and self.m.use_vq and self.m.num_quantizers == 1 and self.defer_ema) else None
        works = []
        for g, action in segs:
            g.replay()
            if action is not None:
                self._act(action, works)
        self._pending_ema = None

    def _train_step(self, x, mask, weights, lr, weight_decay, clip, use_graph=True):
        """One whole training step (experiment.py:453 training_step + Lightning backward/clip/AdamW) on the GPU.
        Steady state = hipGraph replays; results land in self.metrics (device).  The captured graph holds every
        kernel of the step; for world_size > 1 it is split around the RCCL all-reduces.  Buffers and graphs belong to the
        arena of this batch shape, so alternating shapes (ragged real data, validation passes) replay safely."""
        m = self.m
        self.train = True
        self.defer_ema = not (m.use_vq and float(m.usage_entropy_lambda) > 0.0)   # the regulariser reads the refreshed table
        a, xt, ms = self._stage_batch(x, mask)    # length-bucketed arena; padded tail masked out
        x_in = self.augment_input(xt)             # eager, outside the graph (fresh torch random draws every step)
        upd = self._host_prologue()
        world = self.world()
        from .parallel import dp_active
        dp = dp_active()                          # data-parallel form of the step (world > 1)
        self.set_hyper(lr, weight_decay, clip, betas=getattr(self, "betas", (0.9, 0.999)), grad_scale=1.0 / world,
                       L_logical=int(x.shape[1]))
        decay = float(m.quantizer.decay) if m.use_vq else 0.0
        # everything a captured graph bakes in as a kernel argument (the batch shape is the arena)
        key = (ms is not None, tuple(sorted((k, float(v)) for k, v in weights.items())), upd, decay,
               world, dp, self.drop_scale, float(m.quantizer.beta) if m.use_vq else 0.0, float(m.label_smoothing or 0.0),
               x_in is xt, float(m.usage_entropy_lambda), self._soft_vq_key(), self.share_layer0, self.fold_dropout_bwd, self.group_wgrad,
               float(m.xyz_align_alpha), float(m.ss_tv_lambda), m._data_std is not None)
        xs = x_in
        use_graph = bool(use_graph) and os.environ.get("VQH_GRAPH", "1") != "0"
        segs = a.graphs.get(key) if use_graph else None
        if segs is not None:
            a.graphs.move_to_end(key)
            self.last_step_mode = "graph"
            self._replay(segs, dp, upd, decay)
            return self.metrics
        if len(a.seen) > 64:                       # scheduled weights change every epoch: forget keys without a graph
            a.seen = {k: v for k, v in a.seen.items() if k in a.graphs}
        seen = a.seen.get(key, 0)
        a.seen[key] = seen + 1
        # the first step at a (shape, key) runs eager and allocates the arena; the second one captures
        if not (use_graph and seen >= 1):
            self.last_step_mode = "eager"
            self._step_eager(xs, xt, ms, weights, upd, dp)
            return self.metrics
        self.last_step_mode = "capture"
        torch.cuda.synchronize()
        segs = []
        gc_was_on = gc.isenabled()
        gc.disable()        # no cycle collection while a capture is open (torch.cuda.graph collects once on entry): a finaliser
                            # that frees device memory or destroys another graph during capture aborts the process
        try:
            # thread_local: the RCCL watchdog thread may query events while this thread captures
            gen = self._step_gen(xs, xt, ms, weights, upd)
            if not dp:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for _ in gen:
                        pass
                segs.append((g, None))
            else:
                # Record the step segment by segment.  Nothing executes during capture, so the collectives between the
                # segments are skipped here; the replay below runs the step for real, collectives included.
                done = False
                while not done:
                    g = torch.cuda.CUDAGraph()
                    action = None
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        try:
                            action = next(gen)
                        except StopIteration:
                            done = True
                    segs.append((g, action))
        except BaseException:
            # No silent eager fallback: a failure inside capture is a bug in the step and would otherwise hide behind
            # slower launches.  VQH_GRAPH=0 runs without graphs.
            self._pending_ema = None
            a.seen.pop(key, None)
            raise
        finally:
            if gc_was_on:
                gc.enable()
        self._pending_ema = None
        a.graphs[key] = segs
        while len(a.graphs) > MAX_GRAPHS_PER_ARENA:
            a.graphs.popitem(last=False)
        self._replay(segs, dp, upd, decay)
        return self.metrics

    def eval_step(self, x, mask, weights):
        self.train = False
        self.defer_ema = False
        _, xt, ms = self._stage_batch(x, mask)
        upd = self._host_prologue()
        rec, z_e, z_q, idx, stats = self._forward_core(xt, ms, upd)
        self.loss(rec, xt, ms, z_e, z_q, stats, weights, L_logical=int(x.shape[1]))
        return self.metrics

    def advance_rng(self):
        call("vqh_rng_advance", self.rng)

    def metrics_dict(self, weights=None):
        vals = self.metrics.tolist()
        weights = weights if weights is not None else (self.ctx or {}).get("weights", {})
        out = {}
        for k, v in zip(METRIC_KEYS, vals):
            wk = OPTIONAL_METRICS.get(k)
            if wk is not None and not float(weights.get(wk, 0.0)) > 0:
                continue
            out[k] = v
        return out


def _f32(x):
    """Round a python double to the fp32 value torch uses when a tensor is multiplied by a scalar."""
    import struct
    return struct.unpack("f", struct.pack("f", float(x)))[0]
