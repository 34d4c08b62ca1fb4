// Composite entry point: the in-order cache replay of a whole chunk of samples
// (loop body of Uni_Adapter.py:388-521 for --use-mode-dota, per sample) enqueued by ONE host call.
// Pure launch orchestration - it only calls the other extern "C" entry points of this library,
// so semantics are exactly those of the individual kernels; nothing synchronises.
#include "ua_common.h"

extern "C" int ua_tta_chunk(const ua_tta_state* s, const float* feats, const int32_t* targets, int32_t n,
                            int64_t sample_index0, int64_t adam_step0, float* final_logits,
                            int32_t* ranks, int32_t* preds, void* stream) {
    UA_CHECK_PTR(s); UA_CHECK_PTR(feats); UA_CHECK_PTR(final_logits);
    if (n <= 0 || sample_index0 < 0 || adam_step0 < 0) return UA_EINVAL;
    const int K = s->K, M = s->M, D = s->D;
    if (K <= 0 || M <= 0 || D <= 0) return UA_EINVAL;
    if (targets && !ranks) return UA_EINVAL;
    const float* text = s->res_learning ? s->e : s->text0;                       // Uni_Adapter.py:388-401
    int64_t adam_step = adam_step0;
    if (s->res_learning) {
        UA_CHECK_PTR(s->tickets);
        if (hipMemsetAsync(s->tickets, 0, sizeof(int32_t) * K, ua::as_stream(stream)) != hipSuccess) return UA_ELAUNCH;
    }
    for (int j = 0; j < n; ++j) {
        const float* f = feats + (size_t)j * 2 * D;
        const float* fa = f + D;
        int rc;
        if ((rc = ua_clip_logits_pair(f, fa, text, K, D, s->logit_scale, s->fhat, s->faug, s->clip_logits, s->prob,
                                      stream))) return rc;               // :406, :424, :429
        const bool learn = s->res_learning && sample_index0 + j > 0;
        const bool mfma = learn && s->align_mfma && s->t && s->U && s->align_ws && ua_align_mfma_supported(K, M, D);
        if (mfma) {    // residual learning on the matrix cores (csrc/ua_align_mfma.hip)
            if ((rc = ua_gmm_step_fused(s->mu, s->var, s->pi, s->c, s->class_counts, s->fhat, 1, s->fhat, s->faug,
                                        s->prob, s->dota_scores, K, M, D, s->eps, stream))) return rc;                // :416,:417,:430
            if ((rc = ua_align_mfma_prepare(s->mu, s->var, s->pi, K, M, D, s->eps, s->align_ws, s->align_ws_bytes, stream))) return rc;
            for (int it = 0; it < s->res_steps; ++it) {                                                            // :444-476
                if ((rc = ua_align_forward_mfma(s->t, K, M, D, s->align_ws, s->align_ws_bytes, s->L, s->joint, s->U, s->norm, stream))) return rc;
                ++adam_step;
                if (adam_step > 0x7fffffff) return UA_EINVAL;
                if ((rc = ua_align_step_mfma(s->L, s->joint, s->U, s->norm, s->text0, s->res, s->adam_m, s->adam_v, s->t, K, M, D,
                                             (int32_t)adam_step, s->lr, s->beta1, s->beta2, s->adam_eps, s->align_ws, s->align_ws_bytes,
                                             s->loss, nullptr, stream))) return rc;
            }
            if ((rc = ua_text_normalize(s->text0, s->res, K, D, 0.0f, s->e, s->norm, stream))) return rc;             // e = t/|t| for :388-393
        } else if (learn) {   // cache step + the per-sample constants of the residual loop (1/v, row constants) in one launch
            if ((rc = ua_gmm_step_fused_prepare(s->mu, s->var, s->pi, s->c, s->class_counts, s->fhat, 1, s->fhat, s->faug,
                                                s->prob, s->dota_scores, K, M, D, s->eps, s->inv_var, s->row_const, stream))) return rc;
        } else if ((rc = ua_gmm_step_fused(s->mu, s->var, s->pi, s->c, s->class_counts, s->fhat, 1, s->fhat, s->faug,
                                           s->prob, s->dota_scores, K, M, D, s->eps, stream))) return rc;   // :416,:417,:430
        if (learn && !mfma) {                                                   // :444-476
            for (int it = 0; it < s->res_steps; ++it) {
                if ((rc = ua_align_forward(s->mu, s->inv_var, s->row_const, s->e, K, K, M, D, s->L, s->joint, stream))) return rc;
                ++adam_step;
                if (adam_step > 0x7fffffff) return UA_EINVAL;
                // loss + gradient, backward and Adam in one launch (bit-identical to the three entry points)
                if (K > 128 || s->replay_lean) {
                    if ((rc = ua_align_loss(s->L, s->joint, K, M, s->loss, s->grad_ll, stream))) return rc;
                    if ((rc = ua_align_backward(s->mu, s->inv_var, s->e, s->grad_ll, K, K, M, D, s->grad_e, stream))) return rc;
                    if ((rc = ua_residual_adam(s->text0, s->res, s->adam_m, s->adam_v, s->grad_e, s->e, s->norm, K, D,
                                               (int32_t)adam_step, s->lr, s->beta1, s->beta2, s->adam_eps, stream))) return rc;
                } else if ((rc = ua_align_step(s->mu, s->inv_var, s->L, s->joint, s->text0, s->res, s->adam_m, s->adam_v, s->e,
                                        s->norm, K, M, D, (int32_t)adam_step, s->lr, s->beta1, s->beta2, s->adam_eps,
                                        s->loss, nullptr, s->grad_e, s->tickets, stream))) return rc;
            }
        }
        if ((rc = ua_fuse_logits(s->clip_logits, s->dota_scores, s->c, K, M, s->rho, s->eta, 1,
                                 targets ? targets + j : nullptr, final_logits + (size_t)j * K,
                                 targets ? ranks + j : nullptr, preds ? preds + j : nullptr, stream))) return rc;   // :491-521
    }
    return UA_OK;
}
