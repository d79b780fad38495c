// libg2vlm_comm.so: RCCL behind the collective entry points of include/g2vlm_comm.h (one process per GPU, xGMI underneath).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <new>
#include <string.h>
#include "g2vlm_comm.h"

namespace {
struct Comm { ncclComm_t nccl; int world, rank; };
constexpr int ERR_ARG = -22, ERR_IO = -5;
inline int rc(ncclResult_t r) { return r == ncclSuccess ? 0 : ERR_IO; }
}  // namespace

extern "C" int g2v_comm_unique_id(void* id_out) {
  if (!id_out) return ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == G2V_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return ERR_IO;
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int g2v_kv_allgather_init(void** comm_out, int world, int rank, const void* id) {
  if (!comm_out || !id || world < 1 || rank < 0 || rank >= world) return ERR_ARG;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  Comm* c = new (std::nothrow) Comm{nullptr, world, rank};
  if (!c) return -12;
  if (ncclCommInitRank(&c->nccl, world, uid, rank) != ncclSuccess) { delete c; return ERR_IO; }
  *comm_out = c;
  return 0;
}

extern "C" int g2v_kv_allgather_run(void* comm, void* full, int64_t block_bytes, void* stream) {
  Comm* c = (Comm*)comm;
  if (!c || !full || block_bytes <= 0 || (block_bytes & 1)) return ERR_ARG;
  const char* mine = (const char*)full + (size_t)c->rank * block_bytes;
  return rc(ncclAllGather(mine, full, (size_t)(block_bytes / 2), ncclBfloat16, c->nccl, (hipStream_t)stream));
}

extern "C" int g2v_kv_allgather_run2(void* comm, void* k_full, void* v_full, int64_t block_bytes, void* stream) {
  Comm* c = (Comm*)comm;
  if (!c || !k_full || !v_full || block_bytes <= 0 || (block_bytes & 1)) return ERR_ARG;
  const size_t off = (size_t)c->rank * block_bytes, n = (size_t)(block_bytes / 2);
  if (ncclGroupStart() != ncclSuccess) return ERR_IO;
  ncclResult_t r1 = ncclAllGather((const char*)k_full + off, k_full, n, ncclBfloat16, c->nccl, (hipStream_t)stream);
  ncclResult_t r2 = ncclAllGather((const char*)v_full + off, v_full, n, ncclBfloat16, c->nccl, (hipStream_t)stream);
  ncclResult_t r3 = ncclGroupEnd();
  return (r1 == ncclSuccess && r2 == ncclSuccess && r3 == ncclSuccess) ? 0 : ERR_IO;
}

extern "C" int g2v_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream) {
  Comm* c = (Comm*)comm;
  if (!c || !buf || bytes <= 0 || root < 0 || root >= c->world) return ERR_ARG;
  return rc(ncclBroadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->nccl, (hipStream_t)stream));
}

extern "C" int g2v_comm_world(void* comm) { return comm ? ((Comm*)comm)->world : ERR_ARG; }
extern "C" int g2v_comm_rank(void* comm) { return comm ? ((Comm*)comm)->rank : ERR_ARG; }

extern "C" int g2v_kv_allgather_destroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ERR_ARG;
  ncclResult_t r = ncclCommDestroy(c->nccl);
  delete c;
  return rc(r);
}
