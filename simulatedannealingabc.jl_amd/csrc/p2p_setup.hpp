// p2p_setup.hpp -- sabc_comm_p2p_setup (include/sabc_hip.h): the peer-to-peer set-up with the shards AGREEING after every
// step, over the collectives already installed.  Either every shard ends on the peer-to-peer transport or every shard
// ends on the collectives -- never a shard that switched while its peers did not (they would wait out the bound of their
// first exchange before falling back), whatever the host language is.
//
// A template over the backend so that tests/cpu_engine runs the very same sequence where no GPU exists.  BE needs:
// p2p_descriptor(P2PDesc *), p2p_init(const P2PDesc *), p2p_selftest(), p2p_disable(), p2p_forget_export(), gather_buffer(),
// to_backend(), to_host(), error().
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "engine.hpp"
#include "p2p.hpp"

namespace sabc {

// the descriptors of all shards over the collectives (an allgather of raw bytes, carried as doubles)
template <class BE>
int p2p_gather_descriptors(BE *be, Collectives *coll, int world, const P2PDesc &mine, std::vector<P2PDesc> &all, std::string *err) {
  constexpr int64_t kD = (int64_t)(sizeof(P2PDesc) / sizeof(double));
  double *g = be->gather_buffer((int64_t)(world + 1) * kD);
  if (!g) { *err = "out of memory for the descriptor exchange"; return SABC_ERR_HIP; }
  if (be->to_backend(g, reinterpret_cast<const double *>(&mine), kD)) { *err = be->error(); return SABC_ERR_HIP; }
  if (coll->allgather(g, g + kD, kD)) { *err = "allgather of the peer-to-peer descriptors failed (no transport installed?)"; return SABC_ERR_COMM; }
  if (be->to_host(reinterpret_cast<double *>(all.data()), g + kD, (int64_t)world * kD)) { *err = be->error(); return SABC_ERR_HIP; }
  return 0;
}

// true on every shard iff `ok` on every shard (an allreduce of a 0 / 1 flag)
template <class BE>
int p2p_all_agree(BE *be, Collectives *coll, bool ok, bool *all_ok, std::string *err) {
  double *g = be->gather_buffer(8);
  double v = ok ? 0.0 : 1.0;
  if (!g) { *err = "out of memory for the agreement flag"; return SABC_ERR_HIP; }
  if (be->to_backend(g, &v, 1)) { *err = be->error(); return SABC_ERR_HIP; }
  if (coll->allreduce_sum(g, 1)) { *err = "allreduce of the agreement flag failed"; return SABC_ERR_COMM; }
  if (be->to_host(&v, g, 1)) { *err = be->error(); return SABC_ERR_HIP; }
  *all_ok = v == 0.0;
  return 0;
}

// 1: every shard now runs peer to peer; 0: every shard stays on the collectives (*note says why); < 0: the collectives failed
template <class BE>
int p2p_setup_sequence(BE *be, Collectives *coll, const Shard &sh, bool host_mode, std::string *note) {
  note->clear();
  // what every shard knows from its own configuration needs no agreement
  if (sh.world < 2 || sh.world > SABC_P2P_MAX_WORLD) { *note = "the peer-to-peer transport takes 2..8 shards (one node)"; return 0; }
  if (host_mode) { *note = "a host-callback simulator or prior keeps to the collectives"; return 0; }
  if (!coll->usable()) { *note = "sabc_comm_p2p_setup exchanges its descriptors over the installed collectives: install them first"; return SABC_ERR_COMM; }
  std::string why;
  bool all_ok = false;
  auto stay = [&](const char *step) {              // every shard leaves: whatever it had mapped is unmapped, the peers are told
    be->p2p_disable();
    // ... and once EVERY shard has (one more agreement, as a barrier: leaving closes the mappings before it returns) nobody
    // holds a mapping of anybody's memory: what was exported for this set-up can be freed by sabc_destroy without waiting for
    // acknowledgements that a shard which never got to know its peers' host pages could not even read
    bool all_left = false;
    if (p2p_all_agree(be, coll, true, &all_left, note) == 0 && all_left) be->p2p_forget_export();
    *note = std::string("peer-to-peer set-up: ") + step + (why.empty() ? std::string(" failed on another shard") : ": " + why) +
            " -- every shard stays on the collectives";
    return 0;
  };
  // 1. descriptors (a shard that cannot export sends an empty one)
  P2PDesc mine;
  bool ok = be->p2p_descriptor(&mine) == 0;
  if (!ok) { why = be->error(); std::memset(&mine, 0, sizeof(mine)); }
  std::vector<P2PDesc> all((size_t)sh.world);
  if (int rc = p2p_gather_descriptors(be, coll, sh.world, mine, all, note)) return rc;
  for (int r = 0; ok && r < sh.world; ++r)
    if (all[(size_t)r].magic != kP2PMagic) { ok = false; why = "a shard could not export its memory"; }
  if (int rc = p2p_all_agree(be, coll, ok, &all_ok, note)) return rc;
  if (!all_ok) return stay("exporting the shards' memory");
  // 2. map every peer
  ok = be->p2p_init(all.data()) == 0;
  if (!ok) why = be->error();
  if (int rc = p2p_all_agree(be, coll, ok, &all_ok, note)) return rc;
  if (!all_ok) return stay("mapping the peers' memory");      // nobody runs the self-test: it would wait for the shard that failed
  // 3. first contact: the slots, then what the transport reads
  ok = be->p2p_selftest() == 0;
  if (!ok) why = be->error();
  if (int rc = p2p_all_agree(be, coll, ok, &all_ok, note)) return rc;
  if (!all_ok) return stay("self-test");
  return 1;
}

}  // namespace sabc
