"""Mirror of the reference's full-covariance cache `dota.DOTA` (dota.py:19-87) on the
MI355X kernels of csrc/ua_dota.hip.  Same constructor, `fit(x, y)`, `update()`, `predict(X)` and attributes
(`mu`, `c`, `Sigma`, `overall_Sigma`, `Lambda` (fp16), `epsilon`); state lives on the HIP device.

`update()` inverts A = (1-eps) mean_k Sigma_k + eps I (SPD) by blocked Schur-complement inversion: the matrix is
split recursively down to 64 x 64 blocks, which `ua_spd_inverse_small` inverts in LDS; everything else is
`ua_gemm` calls (exact-fp32 MFMA) on sub-blocks, using the symmetry of A and of its inverse so that every
product has the form P @ W^T the GEMM kernel computes:
    P = inv(A11);  T = P A12 = P @ A21^T;  T' = A21 P;  S = A22 - A21 @ T'^T;  R = inv(S)
    B12 = -T @ R^T;  B21 = -R @ T^T;  B11 = P - B12 @ T^T.
The reference calls torch.inverse (LU) in fp32; both agree to fp32 accuracy and `Lambda` is then rounded to fp16.

In the reference's loop this cache is unreachable (`--use-mode-dota` defaults to True and the `use_dota` branch never
assigns `final_logits`, Uni_Adapter.py:409-412); `fuse()` implements the rule of the usage note dota_mixture.py:288-293.
"""
import torch

from . import lib


class DOTA(torch.nn.Module):
    def __init__(self, cfg, input_shape, num_classes, clip_weights, streaming_update_Sigma=True, prior_pre_steps=None, device=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError('uni_adapter_amd.DOTA needs a HIP device (MI355X); no CPU path')
        self.streaming_update_Sigma = bool(streaming_update_Sigma)       # False: means / counts only, Sigma stays sigma * I (dota.py:54)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        D, K = int(input_shape), int(num_classes)
        if D % 64 != 0:
            raise RuntimeError(f'input_shape {D}: shape outside the compiled envelope (multiples of 64)')
        self.input_shape, self.num_classes = D, K
        self.epsilon = cfg['epsilon']
        dev = self.device
        self.mu = clip_weights.T.detach().to(dev, torch.float32).contiguous()                       # :27
        self.c = torch.ones(K, dtype=torch.float32, device=dev)                                     # :28
        self.Sigma = (cfg['sigma'] * torch.eye(D, dtype=torch.float32, device=dev)).repeat(K, 1, 1).contiguous()   # :29
        self.overall_Sigma = torch.mean(self.Sigma, dim=0).contiguous()                             # :30
        # pinverse(sigma * I).half() (:31) = I / sigma, closed form (fp64 -> fp16 like the reference)
        self.Lambda = (torch.eye(D, dtype=torch.float64, device=dev) / float(cfg['sigma'])).half().contiguous()
        self._A = torch.empty(D, D, device=dev); self._inv = torch.empty(D, D, device=dev)
        self._W = torch.empty(D, K, dtype=torch.float16, device=dev)
        self._U = torch.empty(D, D, device=dev); self._X2 = torch.empty(D, D, device=dev); self._Xn = torch.empty(D, D, device=dev)
        self.refine_steps = 3
        self._neg = torch.full((D,), -1.0, device=dev)
        self.update_prior = prior_pre_steps is not None                                             # :32-39
        if self.update_prior:
            self.prior_pre_steps = prior_pre_steps
            self.cum_soft_labels = torch.zeros((1, K), dtype=torch.float32, device=dev)
            self.prior_step = 0
        self._dirty = False

    # -- dota.py:42-66 ---------------------------------------------------------------------------
    @torch.no_grad()
    def fit(self, x, y):
        x = x.detach().to(self.device, torch.float32).contiguous()
        y = y.detach().to(self.device, torch.float32).contiguous()
        if self.update_prior:
            self.cum_soft_labels = self.cum_soft_labels + y
            self.prior_step += 1
        lib.call('ua_dota_fit', lib.ptr(self.Sigma) if self.streaming_update_Sigma else None, lib.ptr(self.mu), lib.ptr(self.c), lib.ptr(x),
                 lib.ptr(y), x.shape[0], self.num_classes, self.input_shape, float(self.epsilon), lib.ptr(self.overall_Sigma), lib.ptr(self._A),
                 lib.stream_ptr())
        self._dirty = self.streaming_update_Sigma

    # -- dota.py:69-72 ---------------------------------------------------------------------------
    @torch.no_grad()
    def update(self):
        D = self.input_shape
        if not self._dirty:                        # A is produced by fit(); before the first fit it is (1-eps) sigma I + eps I
            self._A.copy_((1 - self.epsilon) * self.overall_Sigma + self.epsilon * torch.eye(D, device=self.device))
        self._invert(self._A, self._inv)
        # The explicit block inverse in fp32 is ~cond(A) less accurate than an LU solve (measured on the g10 stream:
        # max error 11 vs 0.02 at |Lambda| = 6.8e3).  Newton-Schulz steps X <- 2X - X A X (two more GEMMs each, on the
        # symmetrised iterate) square the error down to the fp32 floor (0.4 after two steps, 0.1 after three).
        # The product is written to a third buffer (ping-pong): ua_gemm runs one tile per workgroup, so with C aliasing
        # the A operand a finished tile would overwrite columns its row panel's other tiles are still reading.
        X, Xn, U, X2 = self._inv, self._Xn, self._U, self._X2
        for _ in range(self.refine_steps):
            X.copy_(0.5 * (X + X.t()))
            torch.mul(X, 2.0, out=X2)
            self._gemm(X, self._A, U, D, D, D)                               # U = X @ A^T = X A
            self._gemm(X, U, Xn, D, D, D, neg=True, residual=X2)             # Xn = 2X - X @ U^T = 2X - X A X
            X, Xn = Xn, X
        self._inv, self._Xn = X, Xn
        lib.call('ua_to_half', lib.ptr(self._inv), self.Lambda.data_ptr(), self._inv.numel(), lib.stream_ptr())

    def _gemm(self, A, W, C, M, N, K, neg=False, residual=None):
        e = lib.GemmEpilogue()
        if neg:
            if self._neg.numel() < N:
                self._neg = torch.full((N,), -1.0, device=self._neg.device)
            e.scale = self._neg.data_ptr()
        if residual is not None:
            e.residual = residual.data_ptr(); e.ldr = residual.stride(0)
        import ctypes
        lib.call('ua_gemm', A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), C.data_ptr(), C.stride(0), M, N, K, ctypes.byref(e),
                 lib.stream_ptr())

    def _invert(self, A, out):
        """out = inverse of the symmetric positive-definite block A (views into (D,D) buffers, sizes multiples of 64)."""
        n = A.shape[0]
        if n <= 64:
            lib.call('ua_spd_inverse_small', A.data_ptr(), A.stride(0), n, out.data_ptr(), out.stride(0), lib.stream_ptr())
            return
        n1 = (n // 2 + 63) // 64 * 64
        n2 = n - n1
        A21, A22 = A[n1:, :n1], A[n1:, n1:]
        P, B12, B21, R = out[:n1, :n1], out[:n1, n1:], out[n1:, :n1], out[n1:, n1:]
        self._invert(A[:n1, :n1], P)
        dev = A.device
        T = torch.empty(n1, n2, device=dev); Tp = torch.empty(n2, n1, device=dev); S = torch.empty(n2, n2, device=dev)
        self._gemm(P, A21, T, n1, n2, n1)                               # T  = P @ A21^T = P A12
        self._gemm(A21, P, Tp, n2, n1, n1)                              # T' = A21 @ P^T = A21 P
        self._gemm(A21, Tp, S, n2, n2, n1, neg=True, residual=A22)      # S  = A22 - A21 @ T'^T
        self._invert(S, R)
        self._gemm(T, R, B12, n1, n2, n2, neg=True)                     # B12 = -T R
        self._gemm(R, T, B21, n2, n1, n2, neg=True)                     # B21 = -R T^T = -R T'
        self._gemm(B12, T, P, n1, n1, n2, neg=True, residual=P)         # B11 = P - B12 @ T^T   (P + T R T')

    # -- dota.py:75-86 ---------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, X):
        X = X.detach().to(self.device)
        Xh = (X if X.dtype == torch.float16 else X.half()).contiguous()
        B = Xh.shape[0]
        scores = torch.empty(B, self.num_classes, device=self.device)
        lib.call('ua_dota_predict', self.Lambda.data_ptr(), lib.ptr(self.mu), Xh.data_ptr(), B, self.num_classes, self.input_shape,
                 self._W.data_ptr(), lib.ptr(scores), lib.stream_ptr())
        out = scores.half()                                           # values are exactly representable (rounded in the kernel)
        if self.update_prior:                                         # :82-85 (promotes to fp32 like the reference)
            prior = self.cum_soft_labels + (self.prior_pre_steps / self.num_classes)
            prior = prior / (self.prior_pre_steps + self.prior_step)
            out = out + torch.log(prior + 1e-10)
        return out

    @torch.no_grad()
    def fuse(self, clip_logits, dota_logits, batch, rho, eta):
        """final = clip + w * dota with w = clamp(rho * mean(c) / batch, max=eta)  (usage note dota_mixture.py:288-293)."""
        w = torch.clamp(rho * self.c.mean() / batch, max=eta)
        return clip_logits + w * dota_logits.float()


class SpdInverse:
    """Inverse of a symmetric positive-definite matrix of any size on the device: zero-pads to a multiple of 64 with an
    identity block, runs DOTA's blocked Schur-complement inversion (64 x 64 leaves in LDS, everything else `ua_gemm`) and
    `refine_steps` Newton-Schulz steps.  Used by the exact-inverse refinement of the cluster cache (utils/math_utils.py:97-111)."""

    def __init__(self, device, refine_steps=2):
        self.dev, self.refine_steps = device, refine_steps
        self._neg = torch.full((64,), -1.0, device=device)

    _gemm = DOTA._gemm
    _invert = DOTA._invert

    @torch.no_grad()
    def __call__(self, A):
        n = A.shape[0]
        N = (n + 63) // 64 * 64
        Ap = torch.eye(N, device=self.dev)
        Ap[:n, :n] = A
        X = torch.empty(N, N, device=self.dev)
        self._invert(Ap, X)
        U, X2, Xn = (torch.empty(N, N, device=self.dev) for _ in range(3))
        for _ in range(self.refine_steps):
            X.copy_(0.5 * (X + X.t()))
            torch.mul(X, 2.0, out=X2)
            self._gemm(X, Ap, U, N, N, N)
            self._gemm(X, U, Xn, N, N, N, neg=True, residual=X2)              # never in place: see DOTA.update
            X, Xn = Xn, X
        return X[:n, :n]
