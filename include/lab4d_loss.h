/* Per-ray loss epilogue of the Lab4D renderer -- engine/model.py:401-611 (dvr_model.compute_recon_loss, mask_losses,
 * apply_loss_weights, get_mask_balance_wt's result as an input) plus the rendered regularisers that the trainer adds as
 * loss terms (model.py:503-526: reg_eikonal, reg_deform_cyc, reg_delta_skin, reg_skin_entropy, reg_gauss_mask).
 * The reference evaluates ~15 terms with several element-wise launches each and a boolean-indexed mean per term
 * (`v[v > 0].mean()`, model.py:602, a host sync); here one pass over the rays produces every term's sum and count of
 * positive elements, a one-block finish forms the weighted terms, and one pass produces every input gradient.
 *
 * R = M*N rays, frame of ray r = r / N.  All tensors fp32, contiguous; rendered.* are the renderer's outputs (the values
 * a term is differentiated with respect to), target.* the batch.  Terms, in this order (LAB4D_LOSS_TERMS of them):
 *   0 mask          (mask - t_mask)^2 * balance_wt * vis2d * detected[frame]                 (R,1)
 *   1 feature       |feature - t_feature|_2 * t_mask * detected                               (R,16) -> (R,1)
 *   2 feat_reproj   |xy_reproj - t_hxy[:2]|_2 * t_mask * detected, / train_res                (R,2)  -> (R,1)
 *   3 rgb           (rgb - t_rgb)^2 * t_mask * vis2d            element-wise: 3 elements/ray  (R,3)
 *   4 depth         |depth - t_depth| * t_mask * vis2d                                        (R,1)
 *   5 flow          |flow - t_flow|_2 * [t_flow_uct > 0] * t_mask * vis2d, / train_res        (R,2)  -> (R,1)
 *   6 vis           vis * t_mask * vis2d                                                      (R,1)
 *   7 reg_gauss_mask (gauss_mask - stopgrad(mask))^2                                          (R,1)
 *   8 reg_eikonal   eikonal          9 reg_deform_cyc  cyc_dist        10 reg_delta_skin  delta_skin
 *  11 reg_skin_entropy  skin_entropy                                                          (R,1) each
 * loss[k] = weight[k] * scale[k] * sum(v_k * [v_k > 0]) / count(v_k > 0)    (scale = 1/train_res for terms 2 and 5).
 * mask_scale multiplies t_mask in terms 3-6: 1 for field_type "fg"; the comp configuration (vis2d-only masking,
 * model.py:566-571) passes t_mask = NULL there.  Any rendered pointer may be NULL: the term is skipped (loss 0). */
#ifndef LAB4D_LOSS_H
#define LAB4D_LOSS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LAB4D_LOSS_TERMS 12
typedef struct {
  /* rendered */
  const float *mask, *feature, *xy_reproj, *rgb, *depth, *flow, *vis, *gauss_mask, *eikonal, *cyc_dist, *delta_skin, *skin_entropy;
  /* targets */
  const float *t_mask, *t_feature, *t_hxy, *t_rgb, *t_depth, *t_flow, *t_flow_uct, *t_vis2d, *t_detected, *balance_wt;
  int hxy_ld;            /* row stride of t_hxy (3 for homogeneous pixel coordinates) */
  int dense_uses_mask;   /* 1: terms 3-6 are masked by t_mask * vis2d (field_type fg); 0: by vis2d only (comp) */
  /* field_type "comp" (model.py:455-461, 486-493), both optional: `mask` is then the rendered FOREGROUND mask (mask_fg),
   * mask_all the composite's mask -- term 0 becomes ((mask - t_mask)^2 * balance_wt + (mask_all - 1)^2) * vis2d * detected (the
   * composite must be opaque) -- and vis_bg the background field's visibility loss: term 6 becomes (vis + vis_bg_wt * vis_bg) * m. */
  const float *mask_all, *vis_bg;
  float vis_bg_wt;
} lab4d_loss_inputs;
typedef struct {
  float *mask, *feature, *xy_reproj, *rgb, *depth, *flow, *vis, *gauss_mask, *eikonal, *cyc_dist, *delta_skin, *skin_entropy;
  float *mask_all, *vis_bg; /* comp only (may be NULL) */
} lab4d_loss_grads;

/* acc: (2 * LAB4D_LOSS_TERMS) fp32 scratch, zeroed by the call; weights: (LAB4D_LOSS_TERMS) host array (term weight x scale);
 * loss: (LAB4D_LOSS_TERMS + 1) device output, the last entry is the total. */
int lab4d_ray_losses_forward(const lab4d_loss_inputs* in, int R, int N, const float* weights, float* acc, float* loss, void* stream);
/* g_loss: (LAB4D_LOSS_TERMS) device, dL/d loss[k]; acc as left by the forward; every non-NULL pointer of `g` is written. */
int lab4d_ray_losses_backward(const lab4d_loss_inputs* in, int R, int N, const float* weights, const float* acc, const float* g_loss,
                              const lab4d_loss_grads* g, void* stream);

#ifdef __cplusplus
}
#endif
#endif
