/*
 * lab4d_optim.h -- optimizer step of the training loop (included by lab4d_hip.h).  SURVEY.md 8f row 2.
 *
 * Replaces (paths relative to lab4d/):
 *   engine/trainer.py:581-604   check_grad: torch.nn.utils.clip_grad_norm_(param_list, thresh)
 *   engine/trainer.py:164-190   torch.optim.AdamW(one group per parameter, betas (0.9, 0.999), weight_decay 1e-4), stepped at
 *   engine/trainer.py:350       self.optimizer.step()
 * All parameters live in ONE flat fp32 buffer (every parameter padded to a multiple of 4 elements), gradients and the two
 * moments in buffers of the same layout: the step is three launches for the whole model, and the same flat gradient buffer
 * is what a data-parallel job hands to one RCCL all-reduce.
 */
#ifndef LAB4D_OPTIM_H
#define LAB4D_OPTIM_H

/* norm[0] = ||g||_2 over n elements; coef[0] = min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_'s coefficient).  The
 * gradients are not modified: pass `coef` as `grad_scale` to lab4d_adamw_step.  work: >= 512 floats of scratch.  No host sync. */
int lab4d_grad_norm_clip(const float* g, int64_t n, float max_norm, float* work, float* norm, float* coef, void* stream);

/* One AdamW step (decoupled weight decay; torch.optim.AdamW arithmetic) on p, m, v (n elements, n % 4 == 0, 16-byte aligned)
 * with gradient g * grad_scale[0] (grad_scale: device scalar or NULL = 1).  Learning rate of element e = seg_lr[s] for the first
 * segment s with e < seg_end[s] (seg_end: nseg ascending int64 offsets, multiples of 4, last = n; seg_lr: nseg device floats --
 * the per-parameter OneCycleLR rates).  step >= 1 is the 1-based step count of the bias corrections. */
int lab4d_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg,
                     float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale, void* stream);

/* Trainer.check_grad (engine/trainer.py:581-604) without a host round trip: norm / coef as lab4d_grad_norm_clip, plus the reference's
 * discard rule.  skipped[0] = 1 when the pre-clip norm exceeds skip_above (the reference's `if grad_norm > thresh: optimizer.zero_grad()`,
 * after which torch's optimizer skips every parameter: weights, moments and step counts stay as they were) or is not finite (the
 * reference would write NaN into every weight there), else 0.  step[0] (device int32, 0 before the first step) is incremented only
 * when the step is NOT discarded -- it is the 1-based step count lab4d_adamw_step_guarded's bias corrections use.  The caller reads
 * `skipped` whenever it likes (e.g. once per round) to do the reference's cached-weights reload (trainer.py:598-604). */
int lab4d_check_grad(const float* g, int64_t n, float max_norm, float skip_above, float* work, float* norm, float* coef, int32_t* skipped,
                     int32_t* step, void* stream);

/* lab4d_adamw_step with the step count and the discard flag read on the device: a no-op (p, m, v untouched) when skipped[0] != 0,
 * otherwise the AdamW step number step[0] (as left by lab4d_check_grad on the same stream). */
int lab4d_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* seg_lr, int nseg,
                             float beta1, float beta2, float eps, float weight_decay, const float* grad_scale, const int32_t* skipped,
                             const int32_t* step, void* stream);

#endif /* LAB4D_OPTIM_H */
