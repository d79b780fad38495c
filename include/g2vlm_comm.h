/* libg2vlm_comm.so - the collective half of the C-ABI (kv_allgather_{init,run,destroy}, wrapping ncclComm_t).
 *
 * The reference has no multi-GPU inference; these entry points serve the view-sharded prefill of SURVEY 8(e)
 * (g2vlm_amd/sharded.py): per MoT layer every rank writes the K and V rows of its own views into the shared-shape KV cache
 * (replacing the cache merge of reference modeling/g2vlm/qwen2vl.py:626-634) and the ranks all-gather those blocks before the
 * global cross-view attention (reference modeling/g2vlm/g2vlm.py:1030: is_causal=False over ALL views' keys); rank 0 broadcasts
 * view 0's hidden state for the global point decoder (g2vlm.py:1196).
 *
 * One process per GPU.  A separate library from libg2vlm_hip.so so that single-GPU users never load RCCL.  Same conventions as
 * g2vlm_hip.h: extern "C", raw device pointers, the caller's stream, caller-owned buffers, 0 / negative errno-style codes.
 * RCCL calls are stream-ordered: a collective starts after the work already queued on `stream` and the stream continues after
 * it - issue it on a communication stream to overlap it with compute (KVExchange in g2vlm_amd/sharded.py).                      */
#ifndef G2VLM_COMM_H
#define G2VLM_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define G2V_COMM_ID_BYTES 128
/* Rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any side channel
 * (g2vlm_amd/comm.py: torch.distributed's store / a gloo broadcast / a file).                                                  */
int g2v_comm_unique_id(void* id_out);
/* Join the communicator of `world` ranks as `rank` on the CURRENT HIP device.  *comm_out: opaque handle.                        */
int g2v_kv_allgather_init(void** comm_out, int world, int rank, const void* id);
/* In-place all-gather of equal contiguous blocks: `full` holds world * block_bytes bytes, rank r's block at r * block_bytes is
 * valid on rank r before the call and every block is valid on every rank after it (ncclAllGather's in-place form: the send
 * buffer is the rank's own slice of the receive buffer - nothing is staged).  block_bytes % 2 == 0.                            */
int g2v_kv_allgather_run(void* comm, void* full, int64_t block_bytes, void* stream);
/* K and V of one layer as ONE launch (a group of two all-gathers).                                                              */
int g2v_kv_allgather_run2(void* comm, void* k_full, void* v_full, int64_t block_bytes, void* stream);
/* Broadcast `bytes` bytes at `buf` from rank `root` (view 0's context rows, g2vlm.py:1196).                                     */
int g2v_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream);
int g2v_comm_world(void* comm);
int g2v_comm_rank(void* comm);
int g2v_kv_allgather_destroy(void* comm);
#ifdef __cplusplus
}
#endif
#endif
