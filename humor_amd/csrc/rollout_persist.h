// Persistent, weight-stationary HuMoR decoder roll-out (rollout_persist.hip): interface used by rollout.hip.
#pragma once
#include "common.h"

namespace ha {

struct PersistNet;   // register-stationary packing of the decoder + the launch state of one device

// Where one forward call reads its inputs and leaves its results: the same stash regions the launch-chain forward fills
// (one slab per decoder activation: StashLayout::single), so the existing adjoint and the batched prior run unchanged behind it.
struct PersistFwd {
  int B = 0, S = 0;
  const float* past_in0 = nullptr;   // [B][339]
  const float* z_seq = nullptr;      // [B][S][48]
  float* world = nullptr;            // [B][S][348]
  float* xT = nullptr;               // [(S+1)][340][32]   input states of all steps (quad-interleaved 32-row tile)
  float* steps = nullptr;            // per-step region base
  size_t per_step = 0, off_G = 0, off_dec[4] = {0, 0, 0, 0};
  size_t off_gn[3] = {0, 0, 0}, off_gl = 0;   // per step: GroupNorm statistics of the hidden activations; glue record (for the persistent adjoint)
  size_t off_ht[3] = {0, 0, 0};               // per step: the hidden pre-activations in team layout [8][channel][4 rows]
  int dec_pad[4] = {0, 0, 0, 0};              // B > 32 (pipelined kernels): slab widths of the decoder layers, one slab per 32-row tile
  bool hidden_slabs = true;                   // B > 32: also write the hidden pre-activations as launch-chain slabs (for the launch-chain adjoint)
  float* t2j = nullptr;              // [32][3]
  float* ws = nullptr;               // persist_ws_floats() floats of exchange space (zeroed by persist_forward before the launch)
};

// The adjoint of the same roll-out: reads the forward's stash (states, decoder pre-activations, GroupNorm statistics, glue
// records, accumulated transforms) and the prior's dL/dx slabs, writes dL/dpast_in0 and dL/dz.
struct PersistBwd {
  int B = 0, S = 0;
  const float* g_world = nullptr;    // [B][S][348] or null
  const float* gx_pri = nullptr;     // [S][gxp_pad][32] or null
  int gxp_pad = 0;
  const float* xT = nullptr;
  const float* steps = nullptr;
  size_t per_step = 0, off_G = 0, off_dec[4] = {0, 0, 0, 0}, off_gn[3] = {0, 0, 0}, off_gl = 0, off_ht[3] = {0, 0, 0};
  int dec_pad[4] = {0, 0, 0, 0};
  const float* t2j = nullptr;
  float* g_past0 = nullptr;          // [B][339]
  float* g_z = nullptr;              // [B][S][48]
  const float* g_z_add = nullptr;    // [B][S][48] or null: added to g_z by the final reduction (the gradient another reader of z produced)
  float* dz_part = nullptr;          // [S][persist_dz_slots()][32][48]
  float* ws = nullptr;
};

size_t persist_ws_floats();
size_t pipe_ws_floats();      // exchange space of the pipelined kernels (32 < B <= 256)
int pipe_dz_slots();
int persist_dz_slots();      // partial dL/dz products per (step, sequence) written by the persistent adjoint
// *out stays null (and HA_OK is returned) when the network or the device does not have the shape this path is built for
int persist_create(PersistNet** out, int device, const ha_mlp_desc* decoder);
void persist_destroy(PersistNet* p);
// false after a launch has reported a failure (a team that never completed): the caller then uses the launch chain
bool persist_usable(PersistNet* p);
// true exactly once after a persistent launch has reported a failure (the caller returns an error: earlier results are invalid)
bool persist_take_failure(PersistNet* p);
// the caller has observed the failure through the error word and handled it: persist_take_failure stays false from now on
void persist_ack_failure(PersistNet* p);
int persist_forward(PersistNet* p, const PersistFwd& f, int variant, hipStream_t st);
int persist_backward(PersistNet* p, const PersistBwd& f, int variant, hipStream_t st);
// error word of the most recent launches (0 = none); valid after the stream has been synchronised
unsigned persist_error_word(PersistNet* p);
long long persist_launches(PersistNet* p);
long long persist_launches_bwd(PersistNet* p);

}  // namespace ha
