"""Dev check: the torch.distributed calls the N > 1 path makes, on RCCL with a single rank (API / dtype support only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
x = torch.arange(1000, dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
ev = torch.cuda.Event(); ev.record()
with torch.cuda.stream(side):
    side.wait_event(ev)
    dist.all_reduce(x[100:900], op=dist.ReduceOp.AVG)
    done = torch.cuda.Event(); done.record()
torch.cuda.current_stream().wait_event(done)
out = torch.empty(8, 16, device=dev)
dist.all_gather_into_tensor(out, torch.ones(8, 16, device=dev))
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = dist.new_group([0]); c = dist.new_group([0], backend="gloo")
h = torch.zeros(2, dtype=torch.int64); dist.all_reduce(h, op=dist.ReduceOp.MAX, group=c)
from src import dist_utils
k = dist_utils.gather_nograd(torch.ones(4, 8, device=dev))
dist.barrier()
torch.cuda.synchronize()
print("ok", float(x.sum()), tuple(k.shape), float(t))
dist.destroy_process_group()
# the key exchange MoCo runs under data parallelism, on the RCCL backend with one rank: side stream, czc events, two batch
# ranges gathered separately and assembled rank-major, the enqueue ordered behind the main stream, main waits for it
dist.init_process_group("nccl", device_id=dev)
from czech_contriever_amd.parallel import KeyExchange
import czech_contriever_amd.parallel as par
kx = KeyExchange()
kx.world = 1
par_world = par.world
keys = torch.randn(8, 16, device=dev)
got = {}
for n in (4, 0):
    kx._events()
    from czech_contriever_amd import lib as L
    L.check(L.load().czc_event_record(kx.split_event, L.stream_ptr()))
    kx.gather(keys, n)
    kx.enqueue(lambda rows, n=n: got.__setitem__(n, rows.clone()))
torch.cuda.synchronize()
assert torch.equal(got[4], keys) and torch.equal(got[0], keys)
print("key exchange on nccl (1 rank): ok")
dist.destroy_process_group()
