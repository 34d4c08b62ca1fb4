"""The original Uni-Adapter cluster cache with graph-based label refinement on the kernels of
csrc/ua_cluster.hip: `update_cache` / `compute_cache_logits` of Uni_Adapter.py:78-156 and the loop body :541-574, with
`online_value_refinement_new` (conjugate gradients, utils/math_utils.py:21-95).

The reference keeps the cache as a dict {predicted class: [[feature, confidence, prob, count], ...]} and reads `pred`
to the host every sample; the same bookkeeping stays on the host here (slot arithmetic only), all tensors and all
arithmetic live on the device.  `use_new_approximation=False` (:563-566) takes the exact-inverse refinement instead of CG.
In the reference this branch is unreachable (`--use-mode-dota` defaults to True and the branch reads an undefined
`args.use_gmm_dota`, Uni_Adapter.py:541)."""
import ctypes
import math

import torch

from . import lib

LOGIT_SCALE = 100.0


class ClusterCache:
    def __init__(self, text_features, hp, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('uni_adapter_amd.ClusterCache needs a HIP device (MI355X); no CPU path')
        self.dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.text = text_features.detach().to(self.dev, torch.float32).contiguous()        # (K,D)
        self.K, self.D = self.text.shape
        # use_new_approximation False: compute_cache_logits_old / online_value_refinement_old (exact inverse of the regularised
        # Laplacian instead of conjugate gradients; Uni_Adapter.py:159-189, utils/math_utils.py:97-111)
        self.exact_inverse = hp.get('use_new_approximation', True) is False
        self._inv = None
        self.cap, self.beta = int(hp['shot_capacity']), float(hp['beta'])
        self.thr, self.lam = float(hp['threshold']), float(hp['lambda_reg'])
        K, D, cap, dev = self.K, self.D, self.cap, self.dev
        self.keys = torch.zeros(K * cap, D, device=dev); self.prob = torch.zeros(K * cap, K, device=dev)
        self.conf = torch.zeros(K * cap, device=dev); self.count = torch.zeros(K * cap, dtype=torch.int32, device=dev)
        self.sizes = [0] * K
        self._idx = None
        e = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)
        self.fh, self.z, self.p, self.ent = e(1, D), e(1, K), e(1, K), e(1)
        self.pred = e(1, dt=torch.int32); self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.z2, self.final = e(1, K), e(1, K)
        self.last_cg_iterations = 0


    def _cg(self, A, probs_c, P, ld, st):
        K = self.K
        # conjugate gradients on L_reg X = 2 lambda all_probs, vectors transposed (K, ld) (:21-45)
        xT = torch.zeros(K, ld, device=self.dev); rT = torch.zeros(K, ld, device=self.dev); ApT = torch.zeros(K, ld, device=self.dev)
        rT[:, :P] = (2 * self.lam * probs_c).t()
        pT = rT.clone()
        rz = (rT * rT).sum(1).contiguous()
        it = 0
        for it in range(1, 101):
            lib.call('ua_gemm', lib.ptr(pT), ld, lib.ptr(A), ld, lib.ptr(ApT), ld, K, P, ld, None, st)        # (A p)^T = p^T @ A^T
            self.flag.zero_()
            lib.call('ua_cluster_cg_step', lib.ptr(xT), lib.ptr(rT), lib.ptr(pT), lib.ptr(ApT), lib.ptr(rz), K, P, ld, 1e-5,
                     lib.ptr(self.flag, torch.int32), st)
            if not int(self.flag.item()):                                                # `if torch.all(rz_new < tol): break` (:39)
                break
        self.last_cg_iterations = it
        return xT

    @torch.no_grad()
    def step(self, feat):
        """One sample through the loop body (Uni_Adapter.py:406, :541-574).  feat (1,D) encoder output.
        Returns (final_logits (1,K), cache_logits (1,K))."""
        K, D, cap, st = self.K, self.D, self.cap, lib.stream_ptr()
        feat = feat.detach().to(self.dev, torch.float32).contiguous()
        lib.call('ua_clip_logits', lib.ptr(feat), lib.ptr(self.text), K, D, LOGIT_SCALE, 1, lib.ptr(self.fh), lib.ptr(self.z), lib.ptr(self.p),
                 lib.ptr(self.ent), lib.ptr(self.pred, torch.int32), st)
        pred = int(self.pred.item())                                                     # :67 (the reference syncs here too)
        conf_new = torch.exp(-self.beta * (self.ent / math.log2(K)))                     # :80 with get_entropy :28-30
        n = self.sizes[pred]
        if n < cap:                                                                      # :85-88, :116-117
            slot = pred * cap + n
            self.keys[slot].copy_(self.fh[0]); self.prob[slot].copy_(self.p[0]); self.conf[slot] = conf_new[0]; self.count[slot] = 1
            self.sizes[pred] = n + 1
            self._idx = None
        else:                                                                            # :89-114
            o = pred * cap
            lib.call('ua_cluster_merge', self.keys[o:].data_ptr(), self.conf[o:].data_ptr(), self.count[o:].data_ptr(), self.prob[o:].data_ptr(),
                     lib.ptr(self.fh), lib.ptr(conf_new), lib.ptr(self.text), cap, K, D, self.beta, st)
        if self._idx is None:                                                            # sorted(cache.keys()), insertion order within a class
            self._idx = torch.tensor([c * cap + s for c in range(K) for s in range(self.sizes[c])], device=self.dev)
        keys_c = self.keys.index_select(0, self._idx).contiguous()
        probs_c = self.prob.index_select(0, self._idx)
        P = keys_c.shape[0]
        ld = (P + 3) // 4 * 4
        kn = torch.empty(P, D, device=self.dev)
        lib.call('ua_text_normalize', lib.ptr(keys_c), None, P, D, 1e-12, lib.ptr(kn), None, st)              # F.normalize (math_utils.py:50)
        A = torch.zeros(P, ld, device=self.dev)
        lib.call('ua_gemm', lib.ptr(kn), D, lib.ptr(kn), D, lib.ptr(A), ld, P, P, D, None, st)                # W = keys @ keys^T (:51)
        s = torch.empty(P, device=self.dev)
        lib.call('ua_cluster_laplacian', lib.ptr(A), P, ld, self.thr, self.lam, lib.ptr(s), st)               # :52-58
        if self.exact_inverse:
            if self._inv is None:
                from .dota import SpdInverse
                self._inv = SpdInverse(self.dev)
            Li = self._inv(A[:, :P]).contiguous()                                                             # torch.linalg.inv(L_reg) (:107)
            Bt = torch.zeros(K, ld, device=self.dev)
            Bt[:, :P] = (2 * self.lam * probs_c).t()
            Lp = torch.zeros(P, ld, device=self.dev); Lp[:, :P] = Li
            xT = torch.zeros(K, ld, device=self.dev)
            lib.call('ua_gemm', lib.ptr(Bt), ld, lib.ptr(Lp), ld, lib.ptr(xT), ld, K, P, ld, None, st)       # (L_inv @ B)^T = B^T @ L_inv^T (:109)
            self.last_cg_iterations = 0
        else:
            xT = self._cg(A, probs_c, P, ld, st)
        cls = torch.empty(P, dtype=torch.int32, device=self.dev); aff = torch.empty(P, device=self.dev)
        lib.call('ua_cluster_logits', lib.ptr(xT), ld, lib.ptr(keys_c), lib.ptr(self.fh), lib.ptr(self.z), P, K, D, lib.ptr(cls, torch.int32),
                 lib.ptr(aff), lib.ptr(self.z2), lib.ptr(self.final), st)
        return self.final.clone(), self.z2.clone()
