"""Does this torch build's gloo backend take device tensors (broadcast, all_gather_into_tensor)?  Two ranks on cuda:0."""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    x = torch.full((4,), float(rank), device="cuda:0")
    for name, fn in (("broadcast", lambda: dist.broadcast(x, src=1)),
                     ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(8, device="cuda:0"), x)),
                     ("async broadcast", lambda: dist.broadcast(x, src=0, async_op=True).wait())):
        try:
            fn(); torch.cuda.synchronize(); print(rank, name, "ok", x.tolist(), flush=True)
        except Exception as e:
            print(rank, name, "FAILED", type(e).__name__, str(e)[:200], flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(29533,), nprocs=2)
