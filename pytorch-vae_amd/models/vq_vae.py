# -*- coding: utf-8 -*-
"""
Host-side mirror of the reference model interface (models/vq_vae.py of jluuser/PyTorch-VAE) on top
of the MI355X kernels.  Same constructor keywords, attribute names, method names and state_dict
keys, so checkpoints and the reference's scripts switch over unchanged:

    VQVAE(**model_params)                      reference models/vq_vae.py:366-409
    .forward(x, mask) -> [recons, target, (z_q, z_e, idx, ppl, dead), mask]        :767-901
    .loss_function(recons, target, vq_pack, mask, **weights) -> dict               :1097-1388
    .encode / ._tokenize_to_codes / .decode / .generate / .sample                  :639-765, :1390-1422
    .init_codebook_from_centroids(C), .beta, .quantizer.{embedding, ...}           :555-613, :19-283

The modules below are PARAMETER CONTAINERS ONLY: they are created with torch's own initialisers in
the same order as the reference so a given torch seed yields the same initial weights, but none of
their forward() methods is ever used.  All arithmetic runs in vqvae_hip.StepEngine -> libvqvae_hip.so;
there is no CPU path (a missing library or GPU raises vqvae_hip.lib.VqhError).
"""
import copy
import math
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from vqvae_hip import lib as _L
from vqvae_hip.engine import METRIC_KEYS, OPTIONAL_METRICS, StepEngine

Tensor = torch.Tensor
__all__ = ["VQVAE", "VectorQuantizerEMA", "LatentTokenizer"]


# ------------------------------------------------------------------------------------------------
# parameter containers
# ------------------------------------------------------------------------------------------------
class _AttnParams(nn.Module):
    """Packed-projection attention parameters with torch.nn.MultiheadAttention's names and init
    order (out_proj constructed first, then xavier on in_proj, zero biases)."""

    def __init__(self, E: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * E))
        self.out_proj = nn.Linear(E, E)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)


class _EncLayerParams(nn.Module):
    def __init__(self, H: int, ff: int = 2048):
        super().__init__()
        self.self_attn = _AttnParams(H)
        self.linear1 = nn.Linear(H, ff)
        self.linear2 = nn.Linear(ff, H)
        self.norm1 = nn.LayerNorm(H)
        self.norm2 = nn.LayerNorm(H)


class _DecLayerParams(nn.Module):
    def __init__(self, H: int, ff: int = 2048):
        super().__init__()
        self.self_attn = _AttnParams(H)
        self.multihead_attn = _AttnParams(H)
        self.linear1 = nn.Linear(H, ff)
        self.linear2 = nn.Linear(ff, H)
        self.norm1 = nn.LayerNorm(H)
        self.norm2 = nn.LayerNorm(H)
        self.norm3 = nn.LayerNorm(H)


class _Stack(nn.Module):
    """N deep copies of one initialised layer: like nn.TransformerEncoder/Decoder, every layer of a
    stack starts from identical weights."""

    def __init__(self, layer: nn.Module, n: int):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])


class LatentTokenizer(nn.Module):
    """Learnable-query tokenizer parameters, L -> N tokens (reference :288-322)."""

    def __init__(self, d_model: int, n_tokens: int = 32, n_heads: int = 8, n_layers: int = 2, dropout: float = 0.1):
        super().__init__()
        self.n_tokens, self.d = int(n_tokens), int(d_model)
        self.queries = nn.Parameter(torch.randn(self.n_tokens, self.d) * 0.02)
        self.layers = nn.ModuleList()
        for _ in range(int(n_layers)):
            blk = nn.ModuleDict()
            blk["ln_q"] = nn.LayerNorm(self.d)
            blk["ln_kv"] = nn.LayerNorm(self.d)
            blk["attn"] = _AttnParams(self.d)
            blk["ln_o"] = nn.LayerNorm(self.d)
            ffn = nn.ModuleList([nn.Linear(self.d, 4 * self.d), nn.Identity(), nn.Linear(4 * self.d, self.d)])
            blk["ffn"] = ffn
            self.layers.append(blk)


class VectorQuantizerEMA(nn.Module):
    """EMA codebook state + the reference's quantizer API (:19-283). Buffers, not parameters."""

    def __init__(self, num_embeddings: int, embedding_dim: int, beta: float = 0.25, decay: float = 0.98,
                 eps: float = 1e-5, reinit_dead_codes: bool = True, reinit_prob: float = 1.0,
                 dead_usage_threshold: int = 0, print_init: bool = True, diag_qe_cap: float = 10.0,
                 diag_qe_bins: int = 64, num_quantizers: int = 1):
        super().__init__()
        self.num_quantizers = int(num_quantizers)
        self.K_per = int(num_embeddings)
        self.K = self.num_quantizers * self.K_per
        self.D = int(embedding_dim)
        self.beta, self.decay, self.eps = float(beta), float(decay), float(eps)
        self.use_ema = True
        self.reinit_dead_codes = bool(reinit_dead_codes)
        self.reinit_prob = float(reinit_prob)
        self.dead_usage_threshold = int(dead_usage_threshold)
        self.diag_qe_cap, self.diag_qe_bins = float(diag_qe_cap), int(diag_qe_bins)
        self.register_buffer("embedding", torch.randn(self.K, self.D) * (1.0 / math.sqrt(self.D)))
        self.register_buffer("ema_cluster_size", torch.zeros(self.K))
        self.register_buffer("ema_embedding", torch.zeros(self.K, self.D))
        self.register_buffer("_ep_usage", torch.zeros(self.K))
        for n in ("_ep_top1_sum", "_ep_top2_sum", "_ep_cnt", "_ep_qe_sum"):
            self.register_buffer(n, torch.zeros(1))
        self.register_buffer("_ep_qe_hist", torch.zeros(self.diag_qe_bins))
        self._owner = None
        if print_init:
            kind = "RVQ" if self.num_quantizers > 1 else "VQ"
            print(f"[{kind}] EMA (L2) on HIP: levels={self.num_quantizers}, K_per={self.K_per}, K_total={self.K}, "
                  f"D={self.D}, beta={self.beta}, decay={self.decay}")

    @torch.no_grad()
    def reset_epoch_stats(self):
        for n in ("_ep_usage", "_ep_top1_sum", "_ep_top2_sum", "_ep_cnt", "_ep_qe_sum", "_ep_qe_hist"):
            getattr(self, n).zero_()

    @torch.no_grad()
    def get_epoch_stats(self) -> dict:
        """Epoch usage summary (:118-164); margin/qe accumulators are never fed by the reference -> 0."""
        usage = self._ep_usage.detach().cpu()
        cnt = float(self._ep_cnt.item())
        out = {"usage_hist": usage, "margin_mean": 0.0, "qe_mean": 0.0, "qe_p90": 0.0, "n_positions": 0,
               "perplexity": 0.0, "dead_ratio": 0.0}
        if cnt <= 0:
            return out
        out["n_positions"] = int(cnt)
        out["margin_mean"] = float(((self._ep_top1_sum - self._ep_top2_sum) / cnt).item())
        out["qe_mean"] = float((self._ep_qe_sum / cnt).item())
        total = float(usage.sum())
        if total > 0:
            p = (usage / max(total, 1e-12)).clamp_min(1e-12)
            out["perplexity"] = float(torch.exp(-(p * p.log()).sum()))
            out["dead_ratio"] = float((usage == 0).float().mean())
        hist = self._ep_qe_hist.detach().cpu()
        th = float(hist.sum())
        if th > 0:
            cdf = torch.cumsum(hist, 0) / max(th, 1e-12)
            hit = (cdf >= 0.9).nonzero(as_tuple=True)[0]
            i = int(hit[0]) if hit.numel() else self.diag_qe_bins - 1
            out["qe_p90"] = float((i + 0.5) * self.diag_qe_cap / max(self.diag_qe_bins, 1))
        return out

    @torch.no_grad()
    def get_embedding_snapshot(self) -> Tensor:
        return self.embedding.detach().clone()

    def forward(self, z_e: Tensor, do_ema_update: bool = True, allow_reinit: bool = True,
                mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """(z_q_st, z_q, indices, stats) like the reference (:170-283); `mask` [B, M] bool marks the positions that feed
        the EMA statistics and (single level) the usage histogram (:192-205); the model itself never passes one (:869)."""
        if self._owner is None:
            raise _L.VqhError("quantizer is not attached to a VQVAE engine")
        B, M, D = z_e.shape
        eng = self._owner()._engine()
        # a stand-alone call between the model's forward / loss_function / backward must not disturb them: it runs in its
        # own arena and puts the engine's arena, mode flags and pending-refresh state back afterwards
        prev_key, prev = eng.arena.key, (eng.train, eng.defer_ema, eng._pending_ema)
        eng.train = self.training
        eng.defer_ema = False
        eng.use_arena(("vq", int(B), int(M)))
        try:
            with torch.no_grad():
                valid = None if mask is None else mask.to(device=z_e.device, dtype=torch.bool).reshape(B * M)
                z_st, z_q, idx, stats = eng.quantize(z_e.reshape(B * M, D).contiguous(), B, do_ema_update, row_valid=valid)
            idx_out = idx.view(B, M).clone() if self.num_quantizers == 1 else idx.clone()
            out = z_st.view(B, M, D).clone(), z_q.view(B, M, D).clone(), idx_out, stats.clone()
        finally:
            eng.train, eng.defer_ema, eng._pending_ema = prev
            if prev_key in eng.arenas:
                eng.use_arena(prev_key)
        return out


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class VQVAE(nn.Module):
    def __init__(self, input_dim: int = 6, hidden_dim: int = 512, num_layers: int = 4, num_heads: int = 8,
                 max_seq_len: int = 350, codebook_size: int = 512, code_dim: int = 128, beta: float = 0.25,
                 use_vq: bool = True, residual_vq: bool = False, num_quantizers: int = 1,
                 label_smoothing: float = 0.0, ss_tv_lambda: float = 0.0, usage_entropy_lambda: float = 0.0,
                 xyz_align_alpha: float = 0.7, dist_lambda: float = 0.0, rigid_aug_prob: float = 0.0,
                 pairwise_sample_k: int = 32, codebook_init_path: Optional[str] = None,
                 ema_decay_start: float = 0.98, ema_decay_end: float = 0.98, ema_decay_warm_steps: int = 0,
                 soft_vq_use: bool = False, soft_vq_tau_start: float = 2.0, soft_vq_tau_end: float = 0.5,
                 soft_vq_tau_warm_steps: int = 0, soft_vq_alpha_warm_steps: int = 0, noise_warmup_steps: int = 0,
                 max_noise_std: float = 0.0, latent_tokens: int = 32, tokenizer_heads: int = 8,
                 tokenizer_layers: int = 2, tokenizer_dropout: float = 0.1, latent_sigmoid: bool = False,
                 latent_sigmoid_ae_only: bool = True, reinit_dead_codes: bool = True, reinit_prob: float = 1.0,
                 dead_usage_threshold: int = 0, ema_update_freeze_steps: int = 0, print_init: bool = True, **kwargs):
        super().__init__()
        H = int(hidden_dim)
        self.input_dim, self.hidden_dim, self.code_dim = int(input_dim), H, int(code_dim)
        self.max_seq_len = int(max_seq_len)
        self.num_layers, self.num_heads = int(num_layers), int(num_heads)
        self.use_vq = bool(use_vq)
        self._beta = float(beta)
        self.num_quantizers = int(num_quantizers)
        self.residual_vq = self.use_vq and self.num_quantizers > 1
        self.label_smoothing = float(label_smoothing)
        self.ss_tv_lambda = float(ss_tv_lambda)
        self.usage_entropy_lambda = float(usage_entropy_lambda)
        self.xyz_align_alpha = float(xyz_align_alpha)
        self.rigid_aug_prob, self.dist_lambda = float(rigid_aug_prob), float(dist_lambda)
        self.pairwise_sample_k = int(pairwise_sample_k)
        self.ema_decay_start, self.ema_decay_end = float(ema_decay_start), float(ema_decay_end)
        self.ema_decay_warm_steps = int(ema_decay_warm_steps)
        self.soft_vq_use = bool(soft_vq_use)
        self.soft_vq_tau_start, self.soft_vq_tau_end = float(soft_vq_tau_start), float(soft_vq_tau_end)
        self.soft_vq_tau_warm_steps = int(soft_vq_tau_warm_steps)
        self.soft_vq_alpha_warm_steps = int(soft_vq_alpha_warm_steps)
        self.noise_warmup_steps, self.max_noise_std = int(noise_warmup_steps), float(max_noise_std)
        self.codebook_init_path = codebook_init_path
        self.ema_update_freeze_steps = int(ema_update_freeze_steps)
        self._curr_epoch, self.training_steps = 0, 0
        self._ema_decay_override = None
        self._data_mean = self._data_std = None
        if H % self.num_heads or (H // self.num_heads) not in (16, 32, 64):
            raise ValueError("hidden_dim / num_heads must be 16, 32 or 64 for the HIP attention kernels")
        # ---- parameters, created in the reference's order (same seed -> same initial weights) ----
        self.input_proj = nn.Linear(3, H)
        self.ss_input_proj = nn.Linear(3, H)
        self.encoder = _Stack(_EncLayerParams(H), self.num_layers)
        self.enc_ln = nn.LayerNorm(H)
        self.to_code = nn.Linear(H, self.code_dim)
        self.ln_geo = nn.LayerNorm(H)
        self.ln_ss = nn.LayerNorm(H)
        self.ss_encoder = _Stack(_EncLayerParams(H), 2)
        pe = torch.zeros(self.max_seq_len, H)
        pos = torch.arange(0, self.max_seq_len, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, H, 2).float() * (-math.log(10000.0) / H))
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        self.register_buffer("pos_enc", pe.unsqueeze(0))
        self.latent_n_tokens = int(latent_tokens)
        self.tokenizer_heads, self.tokenizer_layers = int(tokenizer_heads), int(tokenizer_layers)
        self.tokenizer_dropout = float(tokenizer_dropout)
        if H % self.tokenizer_heads or (H // self.tokenizer_heads) not in (16, 32, 64):
            raise ValueError("hidden_dim / tokenizer_heads must be 16, 32 or 64")
        self.tokenizer = LatentTokenizer(H, self.latent_n_tokens, self.tokenizer_heads, self.tokenizer_layers,
                                         self.tokenizer_dropout)
        self.latent_sigmoid, self.latent_sigmoid_ae_only = bool(latent_sigmoid), bool(latent_sigmoid_ae_only)
        self.fuse_mlp = nn.ModuleList([nn.Linear(2 * H, H), nn.Identity(), nn.Linear(H, H), nn.LayerNorm(H)])
        if self.use_vq:
            if self.code_dim % 8:
                raise ValueError("code_dim must be a multiple of 8 for the HIP nearest-neighbour kernel")
            self.quantizer = VectorQuantizerEMA(codebook_size, self.code_dim, beta=beta, decay=0.98, eps=1e-5,
                                                reinit_dead_codes=reinit_dead_codes, reinit_prob=reinit_prob,
                                                dead_usage_threshold=dead_usage_threshold, print_init=print_init,
                                                num_quantizers=self.num_quantizers)
            self.quantizer.beta = self._beta
        else:
            self.quantizer = None
        self.from_code = nn.Linear(self.code_dim, H)
        self.mem_ln = nn.LayerNorm(H)
        self.decoder = _Stack(_DecLayerParams(H), self.num_layers)
        self.query_embed = nn.Embedding(self.max_seq_len, H)
        nn.init.normal_(self.query_embed.weight, std=0.02)
        self.head_xyz = nn.Linear(H, 3)
        self.head_ss = nn.Linear(H, 3)
        self._eng = None
        if self.quantizer is not None:
            import weakref
            self.quantizer._owner = weakref.ref(self)
        if self.use_vq and self.codebook_init_path:
            try:
                self.init_codebook_from_centroids(torch.from_numpy(np.load(self.codebook_init_path).astype(np.float32)))
            except Exception as e:  # the reference reports and carries on (:537-545)
                if print_init:
                    print(f"[VQ] Failed to load codebook: {e}")
        if print_init:
            print(f"[Model] VQVAE (MI355X HIP path): H={H}, Dcode={self.code_dim}, use_vq={self.use_vq}, "
                  f"q_levels={self.num_quantizers}, tokensN={self.latent_n_tokens}")

    # ---- small API ------------------------------------------------------------------------------
    @property
    def beta(self):
        return self._beta

    @beta.setter
    def beta(self, value):
        self._beta = float(value)
        if self.quantizer is not None:
            self.quantizer.beta = float(value)

    def set_epoch_context(self, epoch: int, steps_per_epoch: int = 1):
        self._curr_epoch = int(epoch)

    @torch.no_grad()
    def set_data_stats(self, mean_xyz, std_xyz):
        dev = self.head_xyz.weight.device
        self._data_mean = torch.as_tensor(mean_xyz, dtype=torch.float32, device=dev).view(1, 1, 3)
        self._data_std = torch.as_tensor(std_xyz, dtype=torch.float32, device=dev).view(1, 1, 3)

    @torch.no_grad()
    def init_codebook_from_centroids(self, centroids: Tensor):
        """[K,D] or [levels,K_per,D] centroids -> embedding / ema_embedding, ema_cluster_size = 1 (:576-613)."""
        if self.quantizer is None:
            raise ValueError("Quantizer is not initialized.")
        if centroids.dim() == 3:
            if centroids.shape[2] != self.code_dim:
                raise ValueError(f"Centroid D mismatch: expected {self.code_dim}, got {centroids.shape[2]}")
            if centroids.shape[0] * centroids.shape[1] != self.quantizer.K:
                raise ValueError(f"Centroid K mismatch: expected {self.quantizer.K}, "
                                 f"got {centroids.shape[0] * centroids.shape[1]}")
            flat = centroids.reshape(-1, self.code_dim)
        elif centroids.dim() == 2:
            if tuple(centroids.shape) != (self.quantizer.K, self.code_dim):
                raise ValueError(f"Centroid shape mismatch: expected {(self.quantizer.K, self.code_dim)}, "
                                 f"got {tuple(centroids.shape)}")
            flat = centroids
        else:
            raise ValueError(f"Unsupported centroid shape: {tuple(centroids.shape)}")
        q = self.quantizer
        flat = flat.to(device=q.embedding.device, dtype=q.embedding.dtype)
        q.embedding.copy_(flat)
        q.ema_embedding.copy_(flat)
        q.ema_cluster_size.fill_(1.0)
        print(f"[Codebook Init] Loaded centroids with shape {tuple(centroids.shape)}.")

    # ---- engine plumbing ------------------------------------------------------------------------
    def _engine(self) -> StepEngine:
        dev = self.head_xyz.weight.device
        if dev.type != "cuda":
            raise _L.VqhError("VQVAE runs only on an MI355X: call .to('cuda') first (there is no CPU fallback)")
        old = self._eng
        if old is None or old.dev != dev or not old.params_in_sync():
            # parameters were re-homed (.to(), .float(), load into fresh tensors): rebuild the flat buffers, but keep the
            # optimizer state and the dropout counter -- silently restarting AdamW would change the training run
            self._eng = StepEngine(self)
            if old is not None and old.n_flat == self._eng.n_flat and old.offsets == self._eng.offsets:
                self._eng.flat_m.copy_(old.flat_m)
                self._eng.flat_v.copy_(old.flat_v)
                self._eng.opt_step = old.opt_step
                self._eng.rng.copy_(old.rng)
                self._eng.drop_scale = old.drop_scale
        return self._eng

    @staticmethod
    def _prep(x: Tensor, mask: Optional[Tensor]):
        x = x.contiguous().float()
        if mask is not None:
            mask = mask.contiguous()
            if mask.dtype != torch.bool:
                mask = mask.bool()
        return x, mask

    # ---- inference-style entry points (no gradient tape needed by callers) ---------------------------
    @torch.no_grad()
    def encode(self, x, mask=None):
        eng = self._engine()
        x, mask = self._prep(x, mask)
        eng.train, eng.ctx = self.training, {}
        B, Lq, _ = x.shape
        eng.use_arena((int(B), int(Lq)))
        hf, hg, hs = eng.encode(x, mask)
        H = self.hidden_dim
        return hf.view(B, Lq, H).clone(), hg.view(B, Lq, H).clone(), hs.view(B, Lq, H).clone()

    @torch.no_grad()
    def _tokenize_to_codes(self, h_tokens: Tensor, mask: Optional[Tensor]) -> Tensor:
        eng = self._engine()
        B, Lq, H = h_tokens.shape
        eng.train = self.training
        eng.use_arena((int(B), int(Lq)))
        if eng.ctx is None:
            eng.ctx = {}
        eng.ctx.update({"B": B, "L": Lq, "mask": mask})
        hf = eng.T("fuse.out", B * Lq, H)
        hf.copy_(h_tokens.reshape(B * Lq, H))
        z = eng.tokenize(hf, mask.contiguous() if mask is not None else None, B, Lq)
        return z.view(B, self.latent_n_tokens, self.code_dim).clone()

    @torch.no_grad()
    def decode(self, z_for_decode: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        eng = self._engine()
        B = z_for_decode.shape[0]
        Lq = mask.shape[1] if mask is not None else self.max_seq_len
        eng.train = self.training
        eng.use_arena((int(B), int(Lq)))
        if eng.ctx is None:
            eng.ctx = {}
        z = z_for_decode.reshape(-1, self.code_dim).contiguous().float()
        rec = eng.decode(z, mask.contiguous() if mask is not None else None, B, Lq)
        return rec.view(B, Lq, 6).clone()

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, **kwargs) -> List[Tensor]:
        eng = self._engine()
        x, mask = self._prep(x, mask)
        B, Lq, _ = x.shape
        rec, z_e, z_q, idx, stats = eng.forward(x, mask, train=self.training)
        N, D = self.latent_n_tokens, self.code_dim
        if self.use_vq:
            idx_out = idx.view(B, N) if self.num_quantizers == 1 else idx
            ppl, dead = stats[0], stats[1]
        else:
            idx_out = torch.zeros(B, N, dtype=torch.long, device=x.device)
            ppl = dead = torch.zeros((), device=x.device)
        vq_pack = (z_q.view(B, N, D), z_e.view(B, N, D), idx_out, ppl, dead)
        rec_out = rec.view(B, Lq, 6)
        rec_out._vqh_fwd_id = eng.fwd_id           # lets loss_function / backward check they belong to THIS forward
        return [rec_out, x, vq_pack, mask]

    def loss_function(self, *args, **kwargs) -> dict:
        """Metric dict of the reference (:1357-1388).  `loss` carries a backward hook: calling
        loss.backward() runs the HIP backward pass and exposes gradients as param.grad."""
        recons, target, vq_pack = args[0], args[1], args[2]
        mask = args[3] if len(args) > 3 else None
        eng = self._engine()
        z_q, z_e, _idx, ppl, dead = vq_pack
        B, Lq = target.shape[0], target.shape[1]
        fid = getattr(recons, "_vqh_fwd_id", None)
        if fid is not None and fid != eng.fwd_id:
            raise _L.VqhError("loss_function: `recons` comes from an earlier forward; the engine keeps the activations of "
                              "the LAST forward only (call loss_function right after its forward)")
        rec2 = recons.reshape(B * Lq, 6).contiguous()
        stats = None
        if self.use_vq:                             # the perplexity / dead ratio the caller hands in (:1097, :1366-1367)
            stats = eng.T("loss.stats_in", 2)
            stats[0].copy_(torch.as_tensor(ppl, dtype=torch.float32))
            stats[1].copy_(torch.as_tensor(dead, dtype=torch.float32))
        metrics = eng.loss(rec2, target.contiguous(), mask, z_e.reshape(-1, self.code_dim).contiguous(),
                           z_q.reshape(-1, self.code_dim).contiguous(), stats, kwargs)
        eng.ctx["loss_fwd_id"] = fid               # None: tensors did not come from the engine -> no parameter gradients
        vals = metrics.clone()
        out = {}
        for i, k in enumerate(METRIC_KEYS):
            wk = OPTIONAL_METRICS.get(k)
            if wk is not None and not float(kwargs.get(wk, 0.0)) > 0:
                continue
            out[k] = vals[i]
        if torch.is_grad_enabled() and self.training and fid is not None:
            out["loss"] = _LossBridge.apply(self._anchor(), vals[0], self, fid)
        return out

    def _anchor(self):
        if getattr(self, "_anchor_t", None) is None or self._anchor_t.device != self.head_xyz.weight.device:
            self._anchor_t = torch.zeros((), device=self.head_xyz.weight.device, requires_grad=True)
        return self._anchor_t

    # ---- fused step interface used by the harness (experiment.py) -----------------------------------
    def train_step(self, x, mask, weights, lr, weight_decay, clip, betas=(0.9, 0.999), use_graph=True):
        """forward + loss + backward + (RCCL all-reduce) + clip + AdamW + EMA refresh; returns the device metric
        vector (order = vqvae_hip.engine.METRIC_KEYS).  Steady state is a hipGraph replay."""
        eng = self._engine()
        x, mask = self._prep(x, mask)
        eng.betas = tuple(betas)
        # host batches are copied straight into the step's (length-bucketed) input buffers: StepEngine._stage_batch
        return eng.train_step(x, mask, weights, lr, weight_decay, clip, use_graph=use_graph)

    @torch.no_grad()
    def eval_step(self, x, mask, weights):
        eng = self._engine()
        x, mask = self._prep(x, mask)
        return eng.eval_step(x, mask, weights)

    def metric_names(self):
        return list(METRIC_KEYS)

    def metric_sums(self):
        """Device-side running sums of the metric vector over the train steps since reset_metric_sums()."""
        return self._engine().metrics_acc

    def reset_metric_sums(self):
        eng = self._engine()
        _L.call("vqh_memset", eng.metrics_acc, 0, eng.metrics_acc.numel() * 4)

    def optimizer_state(self):
        """AdamW state in torch.optim's state_dict layout (for Lightning-style checkpoints)."""
        eng = self._engine()
        state, names = {}, [n for n, _ in self.named_parameters()]
        for i, n in enumerate(names):
            o, k = eng.offsets[n], eng.P[n].numel()
            state[i] = {"step": torch.tensor(float(eng.opt_step)), "exp_avg": eng.flat_m[o:o + k].view(eng.P[n].shape).clone(),
                        "exp_avg_sq": eng.flat_v[o:o + k].view(eng.P[n].shape).clone()}
        return {"state": state, "param_groups": [{"params": list(range(len(names)))}]}

    def engine_state(self):
        """Engine state that is neither a weight nor an optimizer moment: the dropout counter [seed, step]."""
        return {"rng": self._engine().rng.detach().cpu().clone()}

    def load_engine_state(self, st):
        self._engine().rng.copy_(torch.as_tensor(st["rng"], dtype=torch.int64))

    def load_optimizer_state(self, sd):
        eng = self._engine()
        names = [n for n, _ in self.named_parameters()]
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            o, k = eng.offsets[n], eng.P[n].numel()
            eng.flat_m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            eng.flat_v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            eng.opt_step = int(float(st["step"]))

    def backward(self, grad_scale: float = 1.0):
        """Explicit HIP backward of the last forward + loss_function; fills param.grad (= grad_scale * d loss / d param)."""
        eng = self._engine()
        if eng.ctx is None or eng.ctx.get("loss_fwd_id") != eng.fwd_id:
            raise _L.VqhError("backward: no loss_function result for the engine's last forward (the loss was evaluated on "
                              "other tensors, or another forward ran in between)")
        # torch semantics: gradients ACCUMULATE until they are consumed (engine optimizer step) or dropped
        # (optimizer.zero_grad() -> p.grad None; zero_grad(set_to_none=False) zeroes the flat buffer in place).  The HIP
        # backward overwrites the flat buffer, so a pending gradient is parked and added back.
        probe = next(iter(self.parameters()))
        pending = getattr(eng, "_grads_pending", False) and probe.grad is not None
        parked = eng.flat_g.clone() if pending else None
        eng.backward()
        if float(grad_scale) != 1.0:                # (loss / accum).backward(), loss scaling
            eng.flat_g.mul_(float(grad_scale))
        if parked is not None:
            _L.call("vqh_add", eng.flat_g, parked, eng.flat_g, eng.flat_g.numel())
        eng._grads_pending = True
        eng.attach_grads()
        if getattr(self, "_grad_monitor_enabled", False):
            self._report_grads()

    # ---- host-side gradient diagnostics (models/vq_vae.py:662-734).  The reference installs tensor hooks; here the
    # gradients live in one flat buffer filled by the HIP backward, so the report runs once after backward. -----------
    def enable_grad_monitor(self, enabled: bool = True):
        self._grad_monitor_enabled = bool(enabled)
        print("[Grad Monitor] Enabled" if enabled else "[Grad Monitor] Disabled")

    def _report_grads(self):
        for name, p in self.named_parameters():
            g = p.grad
            if g is None:
                continue
            if not bool(torch.isfinite(g).all()):
                print(f"[GRAD-ERROR] {name}: NaN or Inf detected!")
                continue
            n = float(g.norm())
            if n > 1e-6:
                print(f"[GRAD] {name}: norm={n:.6f}, mean={float(g.mean()):.6f}, std={float(g.std()):.6f}")

    def print_grad_summary(self):
        if not self.training:
            print("[Grad Summary] Model is in eval mode, no gradients")
            return
        # same name tests, in the same order, as the reference's grouping (:704-719)
        groups = [("Geo branch", lambda n: "encoder" in n and "ss_" not in n),
                  ("SS branch", lambda n: "ss_" in n),
                  ("Fusion", lambda n: "fuse" in n),
                  ("VQ", lambda n: "quantizer" in n),
                  ("Decoder", lambda n: "decoder" in n or "head_ss" in n or "head_xyz" in n)]
        count, total, with_grad = [0] * len(groups), [0.0] * len(groups), 0
        for name, p in self.named_parameters():
            if p.grad is None:
                continue
            with_grad += 1
            for i, (_, belongs) in enumerate(groups):
                if belongs(name):
                    count[i] += 1
                    total[i] += float(p.grad.norm())
                    break
        print(f"[Grad Summary] Total params with grad: {with_grad}")
        for (label, _), c, t in zip(groups, count, total):
            if c > 0:
                print(f"  {label}: {c} params, avg_grad_norm={t / max(c, 1):.6f}")

    @torch.no_grad()
    def generate(self, x: Tensor, mask: Optional[Tensor] = None, **kwargs):
        return self.forward(x, mask=mask)[0]

    @torch.no_grad()
    def sample(self, num_samples: int, device, out_len: Optional[int] = None):
        """Random code indices -> decode (:1394-1422)."""
        if not self.use_vq or self.quantizer is None:
            raise RuntimeError("Quantizer is not initialized for sampling.")
        N = self.latent_n_tokens
        L_out = out_len if out_len is not None else self.max_seq_len
        q = self.quantizer
        dev = q.embedding.device
        z = torch.zeros(num_samples, N, self.code_dim, device=dev)
        eng = self._engine()
        for lv in range(q.num_quantizers):
            idx = torch.randint(0, q.K_per, (num_samples * N,), device=dev)
            part = torch.empty(num_samples * N, self.code_dim, device=dev)
            lo = lv * q.K_per
            _L.call("vqh_vq_gather", q.embedding[lo:lo + q.K_per], self.code_dim, idx, 0, None, 0, part, None,
                    num_samples * N, self.code_dim)
            _L.call("vqh_add", z, part, z, z.numel())
        mask = torch.ones(num_samples, L_out, dtype=torch.bool, device=dev)
        return self.decode(z, mask=mask)


class _LossBridge(torch.autograd.Function):
    """Lets `loss_dict['loss'].backward()` (the Lightning-style call) trigger the HIP backward pass."""

    @staticmethod
    def forward(ctx, anchor, loss_value, model, fwd_id):
        ctx.model, ctx.fwd_id = model, fwd_id
        return loss_value.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.model._engine()
        if eng.fwd_id != ctx.fwd_id:
            raise _L.VqhError("loss.backward(): another forward ran since this loss was computed; the engine holds the "
                              "activations of the last forward only")
        ctx.model.backward(grad_scale=float(grad_out))     # upstream gradient: (loss / accum).backward(), loss scaling
        return None, None, None, None
