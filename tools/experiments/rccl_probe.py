"""Probe (GPU box, one device): what RCCL accepts there.
  (a) 1-rank communicator: grouped send-to-self + recv-from-self, broadcast, all_reduce
  (b) two ranks on ONE device: does ncclCommInitRank accept it (RCCL normally refuses duplicate devices)?"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


A = r"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
x = torch.arange(1000, dtype=torch.float32, device='cuda'); y = torch.zeros_like(x)
try:
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, x, 0), dist.P2POp(dist.irecv, y, 0)])
    for r in reqs: r.wait()
    torch.cuda.synchronize()
    print('SELF_SENDRECV', bool(torch.equal(x, y)))
except Exception as e:
    print('SELF_SENDRECV_FAIL', repr(e))
dist.destroy_process_group()
"""

B = r"""
import os, sys, torch, torch.distributed as dist
rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK=str(rank), WORLD_SIZE='2')
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    x = torch.full((4,), float(rank + 1), device='cuda')
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print('TWO_ON_ONE rank', rank, x.tolist())
    dist.destroy_process_group()
except Exception as e:
    print('TWO_ON_ONE_FAIL rank', rank, repr(e)[:300])
"""

if __name__ == "__main__":
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", A % free_port()], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-2000:], r.stderr[-1500:])
    port = free_port()
    for extra in ({}, {"NCCL_DEBUG": "WARN"}):
        ps = [subprocess.Popen([sys.executable, "-c", B % port, str(k)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(env, **extra)) for k in range(2)]
        for p in ps:
            try:
                out, _ = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                p.kill()
                out = "TIMEOUT"
            print(out[-1500:])
        port = free_port()
