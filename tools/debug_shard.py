"""debug: 2 ranks on one GPU (gloo), background split over the ranks vs single GPU"""
import os, sys, socket, warnings
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    warnings.filterwarnings("ignore")
    from co_fusion_amd import facade, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 320, 240
    cam = synth.Camera.scaled(W, H); sc = synth.Scene(n_obj=0)
    kw = dict(max_surfels=1 << 19, conf_global_init=0.5, enable_multiple_models=1, model_spawn_offset=50)
    single = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, **kw)
    par = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, rank=rank, world=world, shard_background=1, **kw)
    par.set_allreduce()
    for t in range(3):
        d, rgb, lab, _ = sc.render(cam, t, noise=True)
        single.process_frame(d, rgb, timestamp=t); par.process_frame(d, rgb, timestamp=t)
        a, b = par.model_info(0), single.model_info(0)
        print(f"rank {rank} frame {t}: pose equal {a['pose'].tobytes() == b['pose'].tobytes()} maxdiff {np.abs(a['pose']-b['pose']).max():.3e} count {a['count']} vs {b['count']} icp stats par {par.model_icp_stats(0)} single {single.model_icp_stats(0)}", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
