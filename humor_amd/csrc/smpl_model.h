// Packed SMPL(+H) model constants, resident in HBM (one copy per device, ~100 MB incl. both posedirs layouts).
#pragma once
#include "common.h"

namespace ha {

constexpr int kMaxSubsets = 8;
constexpr int kMaxJoints = 64;   // one lane per joint in the wave-per-frame kernels
constexpr int kChunk = 64;       // vertices per wave chunk

// The per-frame blend coefficient vector c (length Kfull = NB + 1 + P) is ordered so that the part that is
// non-zero when trailing joints have zero pose is a prefix:
//   c[0..NB-1] = betas,  c[NB] = 1 (template),  c[NB+1+k] = pose_feature[k] = (R_{1+k/9} - I)[k%9]
// and the blend matrix rows follow the same order: shapedirs rows, the template row, posedirs rows.
// v_posed[v, comp] = sum_k c[k] * Pd[k][v][comp]
struct VertexSet {
  int n = 0;          // vertices in the set
  int npad = 0;       // padded to a multiple of kChunk
  int nchunks = 0;
  // wave-per-frame layout: [nchunks][Kfull][3][64]
  float* Pd_v = nullptr;
  // coefficient-major copy for the adjoint: [nchunks][3*64][Kp], Kp = Kfull rounded up to 128
  float* Pd_k = nullptr;
  // skinning weights / joint indices, general form: [nchunks][nnz][64]
  float* w = nullptr;
  int32_t* idx = nullptr;
  // the same weights dense per chunk: [nchunks][64 vertices][64 joints] (adjoint: dL/dA_j = sum_v Wc[v][j] * [g (x) v_posed | g])
  float* Wc = nullptr;
  int32_t* ids = nullptr;   // [n] vertex ids (device); nullptr for slot 0
};

}  // namespace ha

struct ha_smpl_model {
  int device = 0;
  mutable bool gco_lds_attr_set = false;     // dense_gco_kernel's > 64 KB dynamic-LDS attribute (single-group launches only)
  mutable bool fused_lds_attr_set = false;   // pose_blend_skin_kernel's 80 KB dynamic-LDS attribute has been set on this handle's device
  int V = 0, J = 0, NB = 0, P = 0;
  int Kfull = 0;      // NB + 1 + P
  int Kfull_pad = 0;  // even
  int Vpad = 0;       // multiple of 64
  int nnz = 0;        // max skinning influences per vertex
  int depth = 0;      // max tree depth (root = 0)
  // joint constants (device)
  float* Jt = nullptr;        // [J,3]   J_regressor @ v_template
  float* Js = nullptr;        // [NB,3,64] J_regressor @ shapedirs, joint-minor (lane = joint reads are coalesced)
  int32_t* parents = nullptr; // [J], parents[0] = -1
  int32_t* jdepth = nullptr;  // [J]
  int32_t* child_start = nullptr;  // [J+1] CSR of children (deterministic parent-side accumulation in backward)
  int32_t* child_idx = nullptr;    // [J]
  int32_t* anc = nullptr;          // [nrounds][64]: the 2^r-th ancestor of every joint (-1: none) -- pointer-jumping forward chain
  int nrounds = 0;                 // smallest r with 2^r > depth
  // dense (slot 0) extras
  float* Pd_m = nullptr;      // MFMA B-operand layout [Vpad/32][KQ][3][64][4] (KQ = k-pair quads)
  float4* w4 = nullptr;       // [V] (nnz <= 4 fast path)
  uint32_t* idx4 = nullptr;   // [V] 4 x uint8 joint ids
  float* Wd = nullptr;        // [Vpad][64] dense skinning weights (nnz <= 4 models; dense backward, A/B operand of the dense dL/dA product)
  // the skinning weights by joint (CSR over all V vertices; entries of a joint sorted by vertex): the dense backward's dL/dA sum
  int32_t* ja_start = nullptr;   // [J+1]
  int32_t* ja_v = nullptr;       // [E] vertex ids
  float* ja_w = nullptr;         // [E] weights
  int32_t* ja_order = nullptr;   // [J] joints by decreasing list length (dealt round-robin to a block's waves)
  // experiment (ha_tune_set("dense_gA_sparse", 2)): per 64-vertex chunk the joints it touches, in groups of 16 slots, and its weights
  // against those slots (SMPL's vertex order is coherent: 8 joints per chunk on average, 17-18 at most -> one or two groups)
  int32_t* gc_joint = nullptr;   // [Vpad/64][32], -1 = unused slot; null when some chunk touches more than 32 joints
  float* gc_w = nullptr;         // [Vpad/64][2 slot groups][4 k-quarters][16 slots][16]: weight of vertex 64 c + 4 e + kq in slot 16 g + s
  int32_t* gc_ng = nullptr;      // [Vpad/64] slot groups in use (1 or 2)
  ha::VertexSet sets[ha::kMaxSubsets];
  // host copies kept for defining subsets later
  std::string* host_blob = nullptr;  // unused placeholder (keeps struct trivially extendable)
  float* h_Pd = nullptr;      // [Kfull][V][3] host, blend matrix rows in coefficient order
  float* h_w = nullptr;       // [V][nnz] host
  int32_t* h_idx = nullptr;   // [V][nnz] host
};
