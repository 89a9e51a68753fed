// p2p_page.hpp -- the host page of p2p.hpp (P2PHostPage) in POSIX shared memory: created by its owner, opened by name by
// peers in other processes.  Host-only; compiles with g++ and hipcc.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <new>

#include "p2p.hpp"

namespace sabc {

// owner side: a fresh, zeroed page under a process-unique name; nullptr if shared memory is not available
inline P2PHostPage *p2p_page_create(char name_out[48]) {
  static std::atomic<unsigned> counter{0};
  for (int attempt = 0; attempt < 8; ++attempt) {
    std::snprintf(name_out, 48, "/sabc-p2p-%d-%u", (int)getpid(), counter.fetch_add(1) + 1u);
    const int fd = shm_open(name_out, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) continue;                               // (a name left behind by a dead process with this pid)
    if (ftruncate(fd, (off_t)kP2PPageBytes) != 0) { (void)close(fd); (void)shm_unlink(name_out); return nullptr; }
    void *p = mmap(nullptr, kP2PPageBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    (void)close(fd);
    if (p == MAP_FAILED) { (void)shm_unlink(name_out); return nullptr; }
    P2PHostPage *pg = new (p) P2PHostPage();
    pg->gen.store(0); pg->state.store(kP2PNone); pg->cur_parity.store(0);
    for (auto &x : pg->released) x.store(0);
    pg->magic = kP2PPageMagic;
    return pg;
  }
  name_out[0] = 0;
  return nullptr;
}

// peer side (another process): read-only view of a shard's page
inline const P2PHostPage *p2p_page_open(const char *name) {
  if (!name || name[0] != '/' || std::strlen(name) >= 48) return nullptr;
  const int fd = shm_open(name, O_RDONLY, 0);
  if (fd < 0) return nullptr;
  void *p = mmap(nullptr, kP2PPageBytes, PROT_READ, MAP_SHARED, fd, 0);
  (void)close(fd);
  if (p == MAP_FAILED) return nullptr;
  const P2PHostPage *pg = reinterpret_cast<const P2PHostPage *>(p);
  if (pg->magic != kP2PPageMagic) { (void)munmap(p, kP2PPageBytes); return nullptr; }
  return pg;
}

inline void p2p_page_unmap(const P2PHostPage *pg) {
  if (pg) (void)munmap(const_cast<P2PHostPage *>(pg), kP2PPageBytes);
}

// owner side, at destroy: the name goes away now, the memory when the last peer unmaps it
inline void p2p_page_destroy(P2PHostPage *pg, const char *name) {
  if (name && name[0]) (void)shm_unlink(name);
  if (pg) (void)munmap(pg, kP2PPageBytes);
}

}  // namespace sabc
