"""MODE-DOTA cache on MI355X: drop-in for the reference `dota_mixture.DOTA_mix`
(dota_mixture.py:7-274) - same constructor, methods and attributes - with the state resident in
HBM as torch tensors and every state-touching operation a hand-written HIP kernel reached
through the C ABI (include/uniadapter.h).  No CPU path: a HIP device is required.
"""
import torch
from torch import nn

from . import lib


def init_state_cpu(cfg, input_shape, num_classes, clip_weights, num_modes):
    """Host logic of DOTA_mix.__init__ (dota_mixture.py:41-111): returns CPU fp32 tensors
    (mu, var, pi, c, class_counts), epsilon and sigma_init.  `clip_weights` is (D,K)."""
    D, K, M = int(input_shape), int(num_classes), int(num_modes)
    epsilon = cfg.get('epsilon', 0.001)
    sigma_cfg = cfg.get('sigma', 1.0)
    if sigma_cfg >= 0.1:                      # :46-54 auto-correction to 1/D
        sigma_init = 1.0 / D
        print(f"[DOTA-GMM] Warning: sigma={sigma_cfg} is too large for CLIP embeddings. "
              f"Auto-corrected to 1/D = {sigma_init:.5f}")
    else:
        sigma_init = sigma_cfg
    center = clip_weights.detach().t().to('cpu', torch.float32)
    if center.shape != (K, D):
        raise ValueError(f'clip_weights must be (D,K)=({D},{K}), got {tuple(clip_weights.shape)}')
    delta = sigma_init * 0.1                  # :70
    mu = center.unsqueeze(1).repeat(1, M, 1).contiguous()
    for m in range(M):                        # :72-79 one axis-aligned offset per mode
        mu[:, m, m % D] = center[:, m % D] + torch.tensor(delta * (m + 1), dtype=torch.float32)
    scale = torch.tensor([1.0 + 0.05 * m for m in range(M)], dtype=torch.float64).to(torch.float32)
    var = (torch.ones(K, M, D) * sigma_init) * scale.view(1, M, 1)         # :86-93
    var = var.clamp(min=1e-8).contiguous()    # :96
    pi = torch.ones(K, M) / M                 # :102
    c = torch.full((K, M), 1.0 / M, dtype=torch.float32)                   # :106
    class_counts = torch.zeros(K)
    return mu, var, pi, c, class_counts, float(epsilon), float(sigma_init)


class _LogLik(torch.autograd.Function):
    """DOTA_mix._log_likelihood with the gradient w.r.t. x that Uni_Adapter.py:228 relies on."""

    @staticmethod
    def forward(ctx, x, mu, var):
        B, D = x.shape
        K, M, _ = mu.shape
        x = x.contiguous()
        out = torch.empty(B, K, M, device=x.device, dtype=torch.float32)
        ones = torch.ones(K, M, device=x.device, dtype=torch.float32)
        lib.call('ua_gmm_loglik', lib.ptr(mu), lib.ptr(var), lib.ptr(ones), lib.ptr(x), B, K, M, D, 0.0,
                 None, None, lib.ptr(out), lib.stream_ptr())
        ctx.save_for_backward(x, mu, var)
        return out

    @staticmethod
    def backward(ctx, g):
        x, mu, var = ctx.saved_tensors
        B, D = x.shape
        K, M, _ = mu.shape
        gx = torch.empty_like(x)
        lib.call('ua_gmm_loglik_bwd', lib.ptr(mu), lib.ptr(var), lib.ptr(x), lib.ptr(g.contiguous()),
                 B, K, M, D, 0.0, lib.ptr(gx), lib.stream_ptr())
        return gx, None, None


class DOTA_mix(nn.Module):
    """Per-class M-mode diagonal Gaussian mixture with streaming EM (reference DOTA_mix)."""

    def __init__(self, cfg, input_shape, num_classes, clip_weights, num_modes=4,
                 streaming_update_Sigma=True, device=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError('uni_adapter_amd.DOTA_mix needs a HIP device (MI355X); no CPU path')
        if not streaming_update_Sigma:
            raise NotImplementedError('streaming_update_Sigma=False is not on the accelerated path')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.input_shape = int(input_shape)
        self.num_classes = int(num_classes)
        self.num_modes = int(num_modes)
        self.streaming_update_Sigma = True
        self.alpha_max = cfg.get('alpha_max', 0.5)
        mu, var, pi, c, cc, self.epsilon, self.sigma_init = init_state_cpu(
            cfg, input_shape, num_classes, clip_weights, num_modes)
        # state lives in HBM, mutated in place by the kernels
        self.mu = mu.to(self.device)
        self.var = var.to(self.device)
        self.pi = pi.to(self.device)
        self.c = c.to(self.device)
        self.class_counts = cc.to(self.device)
        self.t = 0

    # ---- private (same names as the reference) ----
    def _get_var(self):
        return torch.clamp(self.var + self.epsilon, min=1e-8)

    def _log_likelihood(self, x, mu, var):
        return _LogLik.apply(x.float(), mu.contiguous(), var.contiguous())

    def _dims(self):
        return self.num_classes, self.num_modes, self.input_shape

    # ---- public API ----
    def predict(self, x, source_priors=None):
        """(B,D) any float dtype -> (B,K) fp32 scores (dota_mixture.py:236-267)."""
        x = x.to(self.device).float().contiguous()
        K, M, D = self._dims()
        B = x.shape[0]
        scores = torch.empty(B, K, device=self.device, dtype=torch.float32)
        lib.call('ua_gmm_loglik', lib.ptr(self.mu), lib.ptr(self.var), lib.ptr(self.pi), lib.ptr(x),
                 B, K, M, D, self.epsilon, lib.ptr(scores), None, None, lib.stream_ptr())
        if source_priors is not None:             # :260-265 (never taken by the harness)
            p_est = self.class_counts / (self.class_counts.sum() + 1e-10)
            alpha_t = min(self.alpha_max, self.t / (self.t + 100.0))
            p_k = (1 - alpha_t) * source_priors.to(self.device) + alpha_t * p_est
            return scores + torch.log(p_k + 1e-10).unsqueeze(0)
        return scores

    # ---- MFMA (GEMM-form) similarity for query batches (include/uniadapter.h: ua_gmm_loglik_mfma) ----
    def _mfma_ws(self, B):
        K, M, D = self._dims()
        need = lib.mfma_workspace_bytes(B, K, M, D)
        ws = getattr(self, '_mfma_buf', None)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._mfma_buf, self._mfma_valid = ws, False
        return ws

    def mfma_prepare(self, B=1):
        """Build the state-side MFMA operand (1/v | -2 mu/v as split fp16 pairs, row constants) from the CURRENT state.
        Valid until the next fit(); predict_mfma(..., reuse_operand=True) then skips this pass."""
        K, M, D = self._dims()
        ws = self._mfma_ws(B)
        lib.call('ua_gmm_mfma_operand', lib.ptr(self.mu), lib.ptr(self.var), lib.ptr(self.pi), K, M, D, self.epsilon,
                 ws.data_ptr(), ws.numel(), lib.stream_ptr())
        self._mfma_valid = True

    def predict_mfma(self, x, precision='split', want_joint=False, reuse_operand=False):
        """predict() for a batch of queries as a (B x 2D).(2D x KM) product on the matrix cores.  precision 'split':
        fp32-class (three f16 products per term); 'f16': plain fp16 operands (roofline variant, raw-score error ~0.1).
        Returns scores (B,K) [and joint (B,K,M)]."""
        x = x.to(self.device).float().contiguous()
        K, M, D = self._dims()
        B = x.shape[0]
        ws = self._mfma_ws(B)
        if not (reuse_operand and getattr(self, '_mfma_valid', False)):
            self.mfma_prepare(B)
            ws = self._mfma_buf
        scores = torch.empty(B, K, device=self.device, dtype=torch.float32)
        joint = torch.empty(B, K, M, device=self.device, dtype=torch.float32) if want_joint else None
        lib.call('ua_gmm_loglik_mfma', lib.ptr(x), B, K, M, D, {'split': 3, 'f16': 1}[precision], lib.ptr(scores),
                 lib.ptr(joint) if want_joint else None, ws.data_ptr(), ws.numel(), lib.stream_ptr())
        return (scores, joint) if want_joint else scores

    def joint(self, x):
        """Extension: (scores (B,K), joint (B,K,M)) in one launch (batched alignment forward)."""
        x = x.contiguous()
        K, M, D = self._dims()
        B = x.shape[0]
        scores = torch.empty(B, K, device=self.device, dtype=torch.float32)
        joint = torch.empty(B, K, M, device=self.device, dtype=torch.float32)
        lib.call('ua_gmm_loglik', lib.ptr(self.mu), lib.ptr(self.var), lib.ptr(self.pi), lib.ptr(x),
                 B, K, M, D, self.epsilon, lib.ptr(scores), lib.ptr(joint), None, lib.stream_ptr())
        return scores, joint

    def fit(self, x, gamma_class):
        """Streaming EM update with x (B,D), gamma_class (B,K) (dota_mixture.py:162-234)."""
        x = x.to(self.device).float().contiguous()
        g = gamma_class.to(self.device).float().contiguous()
        K, M, D = self._dims()
        B = x.shape[0]
        lib.call('ua_gmm_fit', lib.ptr(self.mu), lib.ptr(self.var), lib.ptr(self.pi), lib.ptr(self.c),
                 lib.ptr(self.class_counts), lib.ptr(x), lib.ptr(g), B, K, M, D, self.epsilon,
                 lib.stream_ptr())
        self.t += B
        self._mfma_valid = False

    def update(self):
        """No-op kept for API compatibility (dota_mixture.py:269-274)."""
        pass

    def step_fused(self, xq, x1, x2, gamma_class, scores=None, round_query_fp16=True):
        """Extension used by the harness: predict(xq.half()) ; fit(x1,g) ; fit(x2,g) in one pass
        over the state (Uni_Adapter.py:416-430).  All inputs (1,D)/(1,K) fp32 on device."""
        K, M, D = self._dims()
        if scores is None:
            scores = torch.empty(1, K, device=self.device, dtype=torch.float32)
        lib.call('ua_gmm_step_fused', lib.ptr(self.mu), lib.ptr(self.var), lib.ptr(self.pi), lib.ptr(self.c),
                 lib.ptr(self.class_counts), lib.ptr(xq), 1 if round_query_fp16 else 0, lib.ptr(x1), lib.ptr(x2),
                 lib.ptr(gamma_class), lib.ptr(scores), K, M, D, self.epsilon, lib.stream_ptr())
        self.t += 2
        self._mfma_valid = False
        return scores
