/*
 * lab4d_hashgrid.h -- multiresolution hash encoding (included by lab4d_hip.h).
 *
 * BASELINE.json config 5 ("hash-grid (instant-NGP) encoding variant") names it; the reference has NO implementation
 * (lab4d/nnutils/nerf.py:98 is a TODO), so there is no reference interface to replace: the definition follows
 * Mueller et al. 2022, section 3 (restated in lab4d_amd/csrc/hashgrid_math.hpp and oracle/hashgrid_oracle.py) and its parity
 * against the reference is UNPINNED.  It would sit where PosEmbedding does (nnutils/embedding.py:69-125) in front of the basefield.
 *
 * x: (S,3) in [0,1]^3 (clamped); table: (L, 2^log2_T, F) fp32; res: (L) int32 grid resolutions (host-computed
 * floor(N_min * b^l)); out: (S, L*F) level-major.  L <= 32, F <= 8, 4 <= log2_T <= 24.
 */
#ifndef LAB4D_HASHGRID_H
#define LAB4D_HASHGRID_H

int lab4d_hashgrid_forward(const float* x, const float* table, const int32_t* res, int S, int L, int log2_T, int F, float* out,
                           void* stream);
/* The same for a field that lives on the box only (Mueller et al. 2022, section 5.4 / appendix E; lab4d_amd/hashfield.py): a point with a
 * coordinate outside [0,1] gets the ZERO encoding without touching the table (the caller masks that sample's outputs, so its row is never
 * looked at) instead of being clamped onto the boundary cells. */
int lab4d_hashgrid_forward_inside(const float* x, const float* table, const int32_t* res, int S, int L, int log2_T, int F, float* out,
                                  void* stream);
/* g_out (S, L*F) -> g_table (L, 2^log2_T, F) ACCUMULATED with fp32 atomics (zero-fill first; may be NULL) and g_x (S,3) written
 * (may be NULL; the trilinear weights' derivative, piecewise constant in x).  A level whose gradient entries are zero for all 64 samples of a
 * wave is skipped (no vertex arithmetic, no atomics): the masked samples of a box-only field cost their 128-byte gradient row and nothing else. */
int lab4d_hashgrid_backward(const float* x, const float* table, const int32_t* res, const float* g_out, int S, int L, int log2_T, int F,
                            float* g_table, float* g_x, void* stream);

/* Round 6: the table gradient of the HASHED levels through packed 2 x fp16 atomics (F = 2 only).  The fp32 atomics above run at the L2 channels'
 * atomic rate (21 G per second on MI355X whatever the footprint, scope or type: tools/probes/atomic_scope.hip), so the lever is their number: one
 * packed atomic carries both features of a vertex.  Levels l < first_f16_level (the dense, direct-indexed ones: thousands of hits per vertex, runs
 * of equal vertices along a ray combined in the wave) keep their fp32 atomics into g_table; levels l >= first_f16_level (hashed: a handful of hits
 * per vertex and launch) are accumulated in g16 -- (L, 2^log2_T) 32-bit words, two halves per vertex, zero on entry -- at a power-of-two scale
 * taken from the launch's largest |g_out| entry (Instant-NGP accumulates its table gradient in fp16 under a loss scale the same way).
 *   lab4d_hashgrid_absmax: *absmax_bits (zero on entry) = bits of max |g_out[i]|, i < n.
 *   lab4d_hashgrid_backward_f16: as lab4d_hashgrid_backward with the split above.
 *   lab4d_hashgrid_flush_f16: g_table[l][v][f] += fp16(g16[l][v][f]) / scale for l >= first_f16_level; g16 cleared for the next launch. */
int lab4d_hashgrid_absmax(const float* g_out, long n, uint32_t* absmax_bits, void* stream);
int lab4d_hashgrid_backward_f16(const float* x, const float* table, const int32_t* res, const float* g_out, int S, int L, int log2_T,
                                int first_f16_level, float* g_table, uint32_t* g16, const uint32_t* absmax_bits, float* g_x, void* stream);
int lab4d_hashgrid_flush_f16(uint32_t* g16, const uint32_t* absmax_bits, int L, int log2_T, int first_f16_level, float* g_table, void* stream);

#endif /* LAB4D_HASHGRID_H */
