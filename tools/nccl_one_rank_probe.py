"""Developer probe: the multi-rank update path (RCCL all-reduce between the hand-written kernels) in a one-rank nccl
communicator -- the only RCCL rehearsal a one-GPU box allows."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from ppo_car_amd.ppo import PPOConfig, Trainer
cfg = PPOConfig(n_envs=4096, n_steps=64, num_rays=16, track=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tracks", "big_track.json"), batch_size=64, train_iters=4)
tr = Trainer(cfg, device="cuda:0", rank=0, world_size=1)
tr.learner.world_size = 2      # take the multi-rank code path (prepare, K10+K11, all_reduce over RCCL, pc_clip_adam) in a 1-rank communicator
tr.world_size = 1
for _ in range(3):
    s = tr.run_epoch()
torch.cuda.synchronize()
print("ok", s["charts/avg_reward"], s["losses/total_loss"], float(tr.learner.flat_param.abs().sum()))
dist.destroy_process_group()
