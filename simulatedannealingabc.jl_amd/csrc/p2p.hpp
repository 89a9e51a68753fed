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
//
// Nothing here reads freed memory either.  A shard LEAVES the group (sabc_comm_p2p_disable, a failed call, a new set-up,
// sabc_destroy) in this order: its host page says `leaving` -> a `leave` word goes into every peer's slots (their waits for
// this shard give up at once) -> its own stream is drained -> it unmaps every peer -> its host page says which peers'
// memory it has released.  A peer that finds a shard gone -- in its page at the entry of a call, or through the leave word
// inside one -- leaves likewise and carries on over the collectives underneath (or returns SABC_ERR_COMM).  sabc_destroy
// frees what peers had mapped only when every peer's page says `released`; if one does not within the bound, that memory
// is PARKED (kept until the process exits) instead: a reader can meet stale particles, never an unmapped page.
#pragma once
#include <atomic>
#include "sabc_types.hpp"

namespace sabc {

constexpr int kP2PRing = 4;
constexpr int kP2PWords = 2 * kMaxPartials;          // LL words of one shard's row of fused sums

// Slot area of one shard, in 8-byte words:
//   sums   [kP2PRing][kMaxPeers][kP2PWords]   row of shard r for exchange `seq` (ring = seq % kP2PRing)
//   bar    [kP2PRing][kMaxPeers]              shard r has reached barrier `seq`
//   commit [kMaxPeers]                        (call tag << 32 | status) of shard r's last sabc_initialize / sabc_update
//   leave  [kMaxPeers]                        (generation << 32 | 1): shard r has left the group of that generation -- a wait
//                                             for one of its posts gives up at once instead of running into the bound
constexpr int64_t kP2PSumsOff = 0;
constexpr int64_t kP2PBarOff = kP2PSumsOff + (int64_t)kP2PRing * kMaxPeers * kP2PWords;
constexpr int64_t kP2PCommitOff = kP2PBarOff + (int64_t)kP2PRing * kMaxPeers;
constexpr int64_t kP2PLeaveOff = kP2PCommitOff + kMaxPeers;
constexpr int64_t kP2PSlotWords = ((kP2PLeaveOff + kMaxPeers + 15) / 16) * 16;

// Every sequence number that travels in the upper half of a slot word is a TAG: the set-up generation in its upper 12 bits,
// the running number in the lower 20.  Words are compared for equality only and a ring entry is overwritten every
// kP2PRing exchanges, so 20 bits never alias; the generation keeps a word posted by a peer of an EARLIER set-up (a no-wait
// status post that was still in flight when the slots were wiped) from matching a number that started over.
constexpr int kP2PSeqBits = 20;
constexpr uint32_t kP2PSeqMask = (1u << kP2PSeqBits) - 1u;
constexpr uint32_t kP2PMaxGen = (1u << (32 - kP2PSeqBits)) - 1u;
SABC_TYPES_HD inline uint32_t p2p_tag(uint32_t gen, uint32_t seq) { return (gen << kP2PSeqBits) | (seq & kP2PSeqMask); }
SABC_TYPES_HD inline uint32_t p2p_tag_gen(uint32_t tag) { return tag >> kP2PSeqBits; }
// the generation a group agrees on: above every member's own last one (each proposes last + 1; all see all descriptors)
inline uint32_t p2p_agree_gen(const uint32_t *proposals, int world) {
  uint32_t g = 1;
  for (int r = 0; r < world; ++r) if (proposals[r] > g) g = proposals[r];
  return g > kP2PMaxGen ? 1u : g;
}

// kernel argument: where every shard's slot area is mapped in THIS process
struct P2PView {
  uint64_t *slots[kMaxPeers];
  int32_t rank, world;
  uint64_t timeout_ticks;        // of the constant-rate wall clock (s_memrealtime)
};

// One page of HOST memory per shard (POSIX shared memory; peers in other processes shm_open it by the name in the
// descriptor, peers in the same process use the pointer), written by its owner only.  It carries what must stay readable
// after device mappings are gone: whether the shard is still in the group, and which peers' memory it has UNMAPPED -- an
// owner frees (or re-exports) memory that peers had mapped only after every one of them has said so here.  No GPU work is
// needed to read it: sabc_update / sabc_initialize look at their peers' pages before they launch anything.
constexpr uint64_t kP2PPageMagic = 0x5341424350414745ull;   // "SABCPAGE"
enum : uint32_t { kP2PNone = 0, kP2PActive = 1, kP2PLeaving = 2, kP2PGone = 3 };
struct P2PHostPage {
  uint64_t magic;
  std::atomic<uint32_t> gen;                     // generation of the owner's current (or last) set-up
  std::atomic<uint32_t> state;                   // kP2PActive ... for that generation
  std::atomic<uint32_t> released[kMaxPeers];     // released[p] = generation of shard p's memory the owner has unmapped
  std::atomic<uint32_t> cur_parity;              // diagnostics: which of its two population buffers is current
};
constexpr size_t kP2PPageBytes = 4096;
static_assert(sizeof(P2PHostPage) <= kP2PPageBytes, "one page");

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
  // which of ptr_pop[0..1] holds the shard's CURRENT population at set-up time.  The buffers flip on every resample, in
  // step on all shards -- but only in calls that succeed: a failed call may leave the shards on different parities, so a
  // reader indexes a peer's buffers by the OWNER's parity (this + flips since set-up), never by its own
  int32_t cur;
  uint32_t gen_proposal;         // this shard's last generation + 1; the group runs on the maximum
  uint64_t ptr_page;             // the shard's P2PHostPage (same process) ...
  char page_name[48];            // ... and its POSIX shared-memory name (other processes)
  unsigned char pad[112];
};
constexpr uint64_t kP2PMagic = 0x5341424350325032ull;   // "SABCP2P2"
static_assert(sizeof(P2PDesc) == SABC_P2P_DESC_BYTES, "sabc_comm_p2p_descriptor writes SABC_P2P_DESC_BYTES bytes");

}  // namespace sabc
