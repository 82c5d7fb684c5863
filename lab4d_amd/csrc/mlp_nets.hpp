// Compile-time layer tables of the per-sample networks (see include/lab4d_mlp.h).
// Layer shapes follow the reference modules exactly (SURVEY.md 8a notes; verified against
// state_dict shapes): per-frame conditioning columns are removed (folded into pf_bias).
#pragma once
#include "lab4d_mlp.h"

namespace lab4d {

struct LS {
  int ke, kin, mout, relu, pf, add_ext, ext_grad;
};
constexpr int pad32(int x) { return (x + 31) / 32 * 32; }

// nerf.py:99-109,134 : PosEmbedding(3,10) -> CondMLP(D=8,W=256,skips=[4],final_act) -> sdf Linear(256,1)
struct NetFgBase {
  static constexpr int ID = LAB4D_NET_FG_BASE, NL = 10, EMB = 0, NFREQ = 10, CIN = 3, SLOTS = 63, KE = 64, COUT = 1, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 256, 1, 1, 0, 0},  {0, 256, 256, 1, 0, 0, 0}, {0, 256, 256, 1, 0, 0, 0},
                               {0, 256, 256, 1, 0, 0, 0}, {64, 256, 256, 1, 1, 0, 0}, {0, 256, 256, 1, 0, 0, 0},
                               {0, 256, 256, 1, 0, 0, 0}, {0, 256, 256, 1, 0, 0, 0}, {0, 256, 256, 1, 0, 0, 1},
                               {0, 256, 1, 0, 0, 0, 0}};
};
// nerf.py:112-123,135-139,208-213 : PosEmbedding(3,12) -> CondMLP(D=2,W=256,final_act) ; + basefield feature ;
// rgb = Linear(256+32 appr, 128) ReLU Linear(128,3)   (appearance code folded into the per-frame bias)
struct NetFgColor {
  static constexpr int ID = LAB4D_NET_FG_COLOR, NL = 5, EMB = 0, NFREQ = 12, CIN = 3, SLOTS = 75, KE = 96, COUT = 3, AUX3 = 0;
  static constexpr LS L[NL] = {{96, 0, 256, 1, 1, 0, 0}, {0, 256, 256, 1, 0, 0, 0}, {0, 256, 256, 1, 0, 1, 0},
                               {0, 256, 128, 1, 1, 0, 0}, {0, 128, 3, 0, 0, 0, 0}};
};
// visibility.py:39-51 : PosEmbedding(3,10) -> CondMLP(D=2,W=64) -> 1
struct NetVis {
  static constexpr int ID = LAB4D_NET_VIS, NL = 3, EMB = 0, NFREQ = 10, CIN = 3, SLOTS = 63, KE = 64, COUT = 1, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 64, 1, 1, 0, 0}, {0, 64, 64, 1, 0, 0, 0}, {0, 64, 1, 0, 0, 0, 0}};
};
// feature.py:77-84 : PosEmbedding(3,6) -> BaseMLP(D=5,W=128,skips=[4]) -> 16
struct NetFeat {
  static constexpr int ID = LAB4D_NET_FEAT, NL = 6, EMB = 0, NFREQ = 6, CIN = 3, SLOTS = 39, KE = 64, COUT = 16, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 128, 1, 0, 0, 0}, {0, 128, 128, 1, 0, 0, 0}, {0, 128, 128, 1, 0, 0, 0},
                               {0, 128, 128, 1, 0, 0, 0}, {64, 128, 128, 1, 0, 0, 0}, {0, 128, 16, 0, 0, 0, 0}};
};
// skinning.py:70-86 : 3B=75 bone coordinates (+128 time embedding +32 code as per-frame bias) -> 64 -> 64 -> B=25
struct NetSkin {
  static constexpr int ID = LAB4D_NET_SKIN, NL = 3, EMB = 1, NFREQ = 0, CIN = 75, SLOTS = 75, KE = 96, COUT = 25, AUX3 = 0;
  static constexpr LS L[NL] = {{96, 0, 64, 1, 1, 0, 0}, {0, 64, 64, 1, 0, 0, 0}, {0, 64, 25, 0, 0, 0, 0}};
};
// the same network for the 18 joints of skel-human (utils/skel_utils.py:348-349; BASELINE configs[2]): 54 coordinates -> 64 -> 64 -> 18
struct NetSkin18 {
  static constexpr int ID = LAB4D_NET_SKIN18, NL = 3, EMB = 1, NFREQ = 0, CIN = 54, SLOTS = 54, KE = 64, COUT = 18, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 64, 1, 1, 0, 0}, {0, 64, 64, 1, 0, 0, 0}, {0, 64, 18, 0, 0, 0, 0}};
};

// skinning.py:70-124 with linear_1 in per-frame affine form (lab4d_mlp.h, LAB4D_NET_SKIN_A): the 64 "embedding" slots are
// relu(aff[frame][j] . [x; 1]); layers 0 / 1 are linear_2 (64 -> 64) and linear_final (64 -> B)
struct NetSkinA {
  static constexpr int ID = LAB4D_NET_SKIN_A, NL = 2, EMB = 2, NFREQ = 0, CIN = 3, SLOTS = 64, KE = 64, COUT = 25, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 64, 1, 0, 0, 0}, {0, 64, 25, 0, 0, 0, 0}};
};
struct NetSkin18A {
  static constexpr int ID = LAB4D_NET_SKIN18_A, NL = 2, EMB = 2, NFREQ = 0, CIN = 3, SLOTS = 64, KE = 64, COUT = 18, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 64, 1, 0, 0, 0}, {0, 64, 18, 0, 0, 0, 0}};
};

// warping.py:105-170,445-483 : DenseWarp(D=2,W=256) post-warp of ComposedWarp: PosEmbedding(3,6)=39 (+128 time embedding
// +32 instance code as per-frame bias) -> 256 -> 256 -> 3 ; one table serves forward_map and backward_map
struct NetDense {
  static constexpr int ID = LAB4D_NET_DENSE, NL = 3, EMB = 0, NFREQ = 6, CIN = 3, SLOTS = 39, KE = 64, COUT = 3, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 256, 1, 1, 0, 0}, {0, 256, 256, 1, 0, 0, 0}, {0, 256, 3, 0, 0, 0, 0}};
};

// warping.py:37-38,94-141 : fg_motion "dense" -- a bare DenseWarp with the class defaults D=6, W=256 (BaseMLP skips=[4], base.py:30-59):
// PosEmbedding(3,6)=39 (+128 time embedding +32 instance code as per-frame bias) -> 256 x 4 -> [input | 256] -> 256 -> 256 -> 3
struct NetDense6 {
  static constexpr int ID = LAB4D_NET_DENSE6, NL = 7, EMB = 0, NFREQ = 6, CIN = 3, SLOTS = 39, KE = 64, COUT = 3, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 256, 1, 1, 0, 0},  {0, 256, 256, 1, 0, 0, 0}, {0, 256, 256, 1, 0, 0, 0}, {0, 256, 256, 1, 0, 0, 0},
                               {64, 256, 256, 1, 1, 0, 0}, {0, 256, 256, 1, 0, 0, 0}, {0, 256, 3, 0, 0, 0, 0}};
};

// multifields.py:86-93 + nerf.py:60-140 : the background field NeRF(num_freq_xyz=6, num_freq_dir=0, appr_channels=0, D=5, W=128):
// PosEmbedding(3,6) -> CondMLP(D=5,W=128,skips=[4],final_act) -> sdf Linear(128,1)
struct NetBgBase {
  static constexpr int ID = LAB4D_NET_BG_BASE, NL = 7, EMB = 0, NFREQ = 6, CIN = 3, SLOTS = 39, KE = 64, COUT = 1, AUX3 = 0;
  static constexpr LS L[NL] = {{64, 0, 128, 1, 1, 0, 0}, {0, 128, 128, 1, 0, 0, 0}, {0, 128, 128, 1, 0, 0, 0}, {0, 128, 128, 1, 0, 0, 0},
                               {64, 128, 128, 1, 1, 0, 0}, {0, 128, 128, 1, 0, 0, 1}, {0, 128, 1, 0, 0, 0, 0}};
};
// PosEmbedding(3,8) -> CondMLP(D=2,W=128,final_act) ; + basefield feature ; rgb = Linear(128 + 3 view direction, 64) ReLU
// Linear(64,3).  The raw view direction (PosEmbedding(3,0), nerf.py:196) is a second per-sample input: it rides in the
// spare slots 6L+3..6L+5 of the 64-slot embedding block (AUX3) and the rgb layer consumes the block with weights that
// are non-zero on exactly those three slots -- no second embedding path in the kernels.
struct NetBgColor {
  static constexpr int ID = LAB4D_NET_BG_COLOR, NL = 5, EMB = 0, NFREQ = 8, CIN = 3, SLOTS = 54, KE = 64, COUT = 3, AUX3 = 1;
  static constexpr LS L[NL] = {{64, 0, 128, 1, 1, 0, 0}, {0, 128, 128, 1, 0, 0, 0}, {0, 128, 128, 1, 0, 1, 0}, {64, 128, 64, 1, 0, 0, 0},
                               {0, 64, 3, 0, 0, 0, 0}};
};

// Hash-grid field (BASELINE config 5; the reference has no such field: nerf.py:98 is a TODO).  Shapes follow Mueller et al. 2022, section 5.4
// (NeRF): a density / geometry net with one hidden layer of 64 on the L*F = 32 hash features, 16 outputs (here: sdf + 15 geometry
// features), and a colour net with two hidden layers of 64 on [16 geometry outputs | view direction].
struct NetHashGeo {
  static constexpr int ID = LAB4D_NET_HASH_GEO, NL = 2, EMB = 1, NFREQ = 0, CIN = 32, SLOTS = 32, KE = 32, COUT = 16, AUX3 = 0;
  static constexpr LS L[NL] = {{32, 0, 64, 1, 0, 0, 0}, {0, 64, 16, 0, 0, 0, 0}};
};
struct NetHashColor {
  static constexpr int ID = LAB4D_NET_HASH_COLOR, NL = 3, EMB = 1, NFREQ = 0, CIN = 19, SLOTS = 19, KE = 32, COUT = 3, AUX3 = 0;
  static constexpr LS L[NL] = {{32, 0, 64, 1, 0, 0, 0}, {0, 64, 64, 1, 0, 0, 0}, {0, 64, 3, 0, 0, 0, 0}};
};

template <class Net>
constexpr int net_wmax() {
  int w = 32;
  for (int l = 0; l < Net::NL; ++l) {
    if (Net::L[l].kin > w) w = Net::L[l].kin;
    if (pad32(Net::L[l].mout) > w) w = pad32(Net::L[l].mout);
  }
  return w;
}

template <class Net>
inline void fill_desc(lab4d_mlp_desc* d) {
  d->n_layers = Net::NL;
  d->emb_kind = Net::EMB;
  d->n_freq = Net::NFREQ;
  d->c_in = Net::CIN;
  d->emb_slots = Net::SLOTS;
  d->ke = Net::KE;
  d->c_out = Net::COUT;
  for (int l = 0; l < Net::NL; ++l) {
    const LS& s = Net::L[l];
    d->layers[l] = {s.ke, s.kin, s.mout, pad32(s.mout), s.relu, s.pf, s.add_ext, s.ext_grad};
  }
}

}  // namespace lab4d
