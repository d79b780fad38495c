"""ctypes binding of libg2vlm_comm.so (include/g2vlm_comm.h) and `RcclComm`, a communicator for g2vlm_amd.sharded that talks to
RCCL through the C-ABI directly - no torch.distributed process group on the data path.

    comm = RcclComm.from_env()                       # torchrun environment: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    res = recon_view_sharded(model, comm, tokenizer, new_token_ids, images)

The 128-byte rendezvous id travels from rank 0 to the others through a torch.distributed TCPStore (a side channel, host only).
The collectives run on the CURRENT torch stream, so KVExchange's side-stream overlap applies as with any stream-ordered
communicator (`overlappable = True`).  RCCL over xGMI has not run on hardware in this pipeline (one GPU per box): world = 1 is
what the GPU tests exercise (library load, communicator init, in-place all-gather, broadcast, destroy on a real device)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libg2vlm_comm.so")
ID_BYTES = 128
_P, _I, _L = C.c_void_p, C.c_int, C.c_int64
_SIGS = {
    "g2v_comm_unique_id": ([_P], _I),
    "g2v_kv_allgather_init": ([C.POINTER(_P), _I, _I, _P], _I),
    "g2v_kv_allgather_run": ([_P, _P, _L, _P], _I),
    "g2v_kv_allgather_run2": ([_P, _P, _P, _L, _P], _I),
    "g2v_comm_broadcast": ([_P, _P, _L, _I, _P], _I),
    "g2v_comm_world": ([_P], _I),
    "g2v_comm_rank": ([_P], _I),
    "g2v_kv_allgather_destroy": ([_P], _I),
}
EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m g2vlm_amd.build` (it links RCCL from /opt/rocm/lib)")
        _lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, res
    return _lib


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def _stream(dev_index):
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(dev_index))


class RcclComm:
    """The Comm interface of g2vlm_amd.sharded (all_gather_blocks / broadcast / barrier) on libg2vlm_comm.so."""
    overlappable = True

    def __init__(self, world, rank, unique_id, device):
        self.device = torch.device(device)
        assert self.device.type == "cuda" and len(unique_id) == ID_BYTES
        torch.cuda.set_device(self.device)
        h = _P()
        buf = (C.c_char * ID_BYTES).from_buffer_copy(bytes(unique_id))
        _ck(lib().g2v_kv_allgather_init(C.byref(h), int(world), int(rank), buf), "g2v_kv_allgather_init")
        self._h, self.world, self.rank = h, int(world), int(rank)

    @staticmethod
    def unique_id():
        buf = (C.c_char * ID_BYTES)()
        _ck(lib().g2v_comm_unique_id(buf), "g2v_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_env(cls, device=None):
        """One process per GPU as `python -m torch.distributed.run` starts them; the id goes through a TCPStore on MASTER_ADDR."""
        import datetime
        from torch.distributed import TCPStore
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        device = device if device is not None else torch.device("cuda", local)
        if world == 1:
            return cls(1, 0, cls.unique_id(), device)
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 1, world, rank == 0,
                         timeout=datetime.timedelta(seconds=120))
        if rank == 0:
            store.set("g2v_comm_id", cls.unique_id())
        return cls(world, rank, store.get("g2v_comm_id"), device)

    def all_gather_blocks(self, full, block_rows):
        """In-place all-gather of rank r's rows [r * block_rows, (r + 1) * block_rows) of `full` (contiguous rows)."""
        assert full.is_cuda and full.is_contiguous() and full.shape[0] == self.world * block_rows
        nbytes = block_rows * full.stride(0) * full.element_size()
        _ck(lib().g2v_kv_allgather_run(self._h, C.c_void_p(full.data_ptr()), nbytes, _stream(self.device.index)), "g2v_kv_allgather_run")

    def all_gather_kv(self, k_full, v_full, block_rows):
        """K and V of one layer as one grouped launch."""
        assert k_full.is_contiguous() and v_full.is_contiguous() and k_full.shape == v_full.shape
        nbytes = block_rows * k_full.stride(0) * k_full.element_size()
        _ck(lib().g2v_kv_allgather_run2(self._h, C.c_void_p(k_full.data_ptr()), C.c_void_p(v_full.data_ptr()), nbytes,
                                        _stream(self.device.index)), "g2v_kv_allgather_run2")

    def broadcast(self, t, src):
        assert t.is_cuda and t.is_contiguous()
        _ck(lib().g2v_comm_broadcast(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(src), _stream(self.device.index)),
            "g2v_comm_broadcast")

    def barrier(self):
        """Every rank has reached this point: a 1-word all-gather (a collective completes on a rank only when all have joined),
        then the host waits for it."""
        flag = torch.zeros((self.world, 2), dtype=torch.int32, device=self.device)
        self.all_gather_blocks(flag, 1)
        torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        if self._h:
            _ck(lib().g2v_kv_allgather_destroy(self._h), "g2v_kv_allgather_destroy")
            self._h = None
