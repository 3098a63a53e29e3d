"""Does GraphedStep recover when capturing the collective fails?  Two gloo ranks on one GPU: gloo collectives cannot be captured,
TSGNN_ONE_GRAPH_ANY_BACKEND=1 makes the auto mode try anyway; the step must fall back to two graphs and still train."""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp_
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(rank, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSGNN_ONE_GRAPH_ANY_BACKEND="1", TSGNN_GRAPH_ALLREDUCE="auto")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    torch.cuda.set_device(0)
    from two_stage_gnn_amd import dense_encoders as E, synthetic
    from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
    class A: bias = True
    torch.manual_seed(1)
    m = E.GcnEncoderGraph(89, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").cuda()
    hb = synthetic.host_batch(seed=rank, B=6, shape="DD", nmax=400)
    g, x, label = synthetic.to_device(hb, torch.device("cuda"))
    tr = FlatTrainer(m, lr=1e-3, clip=2.0)
    gs = GraphedStep(tr, lambda: m.loss(m(x, g)[1], label), warmup=2)
    for _ in range(3):
        gs.step()
    torch.cuda.synchronize()
    print("rank", rank, "mode:", gs.describe(), "steps", float(tr.state[0]), "loss", float(gs.loss), flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp_.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 29650)) for r in range(2)]
    [p.start() for p in ps]
    [p.join(120) for p in ps]
    print("exit codes", [p.exitcode for p in ps])
    for p in ps:
        if p.is_alive():
            p.kill()
