"""Probe: does RCCL on this box accept two ranks on ONE GPU?  (NCCL refuses duplicate GPUs; if RCCL does not, the
in-library RCCL path can be exercised with real multi-rank communicators on the one-GPU box.)"""
import os, sys, subprocess, time

def worker():
    import torch, torch.distributed as dist
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
    t = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: allreduce -> {t.tolist()}", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    if "RANK" in os.environ:
        worker()
        sys.exit(0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    t0 = time.time()
    while time.time() - t0 < 90 and any(p.poll() is None for p in ps):
        time.sleep(1)
    for p in ps:
        if p.poll() is None:
            p.kill()
            print("killed a rank that was still running", flush=True)
    print("exit codes", [p.returncode for p in ps])
