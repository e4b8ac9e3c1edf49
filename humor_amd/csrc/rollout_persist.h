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
  float* t2j = nullptr;              // [32][3]
  float* ws = nullptr;               // persist_ws_floats() floats of exchange space (zeroed by persist_forward before the launch)
};

size_t persist_ws_floats();
// *out stays null (and HA_OK is returned) when the network or the device does not have the shape this path is built for
int persist_create(PersistNet** out, int device, const ha_mlp_desc* decoder);
void persist_destroy(PersistNet* p);
// false after a launch has reported a failure (a team that never completed): the caller then uses the launch chain
bool persist_usable(PersistNet* p);
int persist_forward(PersistNet* p, const PersistFwd& f, int variant, hipStream_t st);
// error word of the most recent launches (0 = none); valid after the stream has been synchronised
unsigned persist_error_word(PersistNet* p);
long long persist_launches(PersistNet* p);

}  // namespace ha
