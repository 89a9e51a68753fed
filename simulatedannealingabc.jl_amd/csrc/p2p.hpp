// p2p.hpp -- the shards of one node exchange through each other's HBM instead of through a collective library.
//
// Why: at the north-star size (n = 1e6 over 8 GPUs) a shard's population update is a ~30 us kernel followed by a sum of
// <= 61 doubles over the shards (n_accept, sum u, sum rho, the moment sums: SimulatedAnnealingABC.jl:334,348-354).  As
// k_reduce_partials -> ncclAllReduce -> k_control that is three dependent launches plus a collective's kernel per update;
// the chain, not the update kernel, sets the 8-GPU rate.  Here every shard owns a small SLOT AREA in fine-grained device
// memory that all peers have mapped (same process: the pointer itself; another process: hipIpcOpenMemHandle), and ONE
// launch does  sum the partial rows -> store the row into every peer's slots -> wait for every peer's row (bounded spin)
// -> add the rows in rank order (bitwise the same sum on every shard) -> the control step.
//
// The row travels in the LL form RCCL uses for small messages: each 8-byte word = (sequence number << 32 | half a
// double), written with ONE 8-byte store, so a reader that sees the sequence number has the data -- no fence between
// payload and flag, one trip over xGMI.  A ring of kP2PRing entries indexed by the sequence number keeps a fast shard
// from overwriting what a slow one has not read yet (a shard can be at most one exchange ahead of a peer's post).
//
// The populations themselves (and rho) are mapped the same way: DifferentialEvolution / StretchMove partners
// (proposals.jl:105-106,141) and the rows a resample draws (:129-132) are read from their owner's HBM, with a flag
// barrier where one shard's kernel reads what another shard's kernel wrote (k_p2p_barrier).
//
// Nothing here can hang: every wait is bounded by P2PView::timeout_ticks; a shard that gives up sets
// ControlBlock::error = SABC_ERR_COMM and ControlBlock::halt, and sabc_update returns per its error contract.
#pragma once
#include "sabc_types.hpp"

namespace sabc {

constexpr int kP2PRing = 4;
constexpr int kP2PWords = 2 * kMaxPartials;          // LL words of one shard's row of fused sums

// Slot area of one shard, in 8-byte words:
//   sums   [kP2PRing][kMaxPeers][kP2PWords]   row of shard r for exchange `seq` (ring = seq % kP2PRing)
//   bar    [kP2PRing][kMaxPeers]              shard r has reached barrier `seq`
//   commit [kMaxPeers]                        (call number << 8 | status) of shard r's last sabc_initialize / sabc_update
constexpr int64_t kP2PSumsOff = 0;
constexpr int64_t kP2PBarOff = kP2PSumsOff + (int64_t)kP2PRing * kMaxPeers * kP2PWords;
constexpr int64_t kP2PCommitOff = kP2PBarOff + (int64_t)kP2PRing * kMaxPeers;
constexpr int64_t kP2PSlotWords = ((kP2PCommitOff + kMaxPeers + 15) / 16) * 16;

// kernel argument: where every shard's slot area is mapped in THIS process
struct P2PView {
  uint64_t *slots[kMaxPeers];
  int32_t rank, world;
  uint64_t timeout_ticks;        // of the constant-rate wall clock (s_memrealtime)
};

// What one shard tells the others so that they can map its memory (sabc_comm_p2p_descriptor); plain bytes, exchanged by
// the caller over whatever it has (torch.distributed, MPI, a list in the same process) or by the library over the
// installed collectives.  Raw pointers are used when exporter and importer are the same process, the IPC handles otherwise.
struct P2PDesc {
  uint64_t magic;
  int32_t pid, device, rank, world;
  int64_t cap, n_global;
  int32_t d, s;
  uint64_t ptr_slots, ptr_pop[2], ptr_rho;
  unsigned char ipc_slots[64], ipc_pop[2][64], ipc_rho[64];
  unsigned char pad[48];
};
constexpr uint64_t kP2PMagic = 0x5341424350325031ull;   // "SABCP2P1"
static_assert(sizeof(P2PDesc) == SABC_P2P_DESC_BYTES, "sabc_comm_p2p_descriptor writes SABC_P2P_DESC_BYTES bytes");

}  // namespace sabc
