"""GPU tests of the drop-in surface around the hot path: the train.py CLI end to end (reference train.py:72-92,280-292,301),
checkpoint / resume on every update path, the gymnasium-shaped attributes train.py reads (train.py:141-142) and the info keys
(car_env.py:599-603), and the multi-rank update with the gradient all-reduce captured inside the epoch graph (RCCL)."""
import json
import os
import socket

import numpy as np
import pytest
import torch

import oracle
import ppo_car_amd as pc
from ppo_car_amd.ppo import PPOConfig, Trainer
from conftest import ROOT, TRACKS

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------
# train.py as a user runs it
# ------------------------------------------------------------------------------------------------
def _run_dirs(out_dir):
    ck = sorted(os.listdir(os.path.join(out_dir, "checkpoints")))
    lg = sorted(os.listdir(os.path.join(out_dir, "logs")))
    return ck, lg


def test_train_cli_lazy_logging_writes_the_same_rows(tmp_path):
    """--lazy-logging prints and logs one epoch behind the device: the same rows as the default order (same seed), every epoch present."""
    import train
    rows = {}
    for name, extra in (("sync", []), ("lazy", ["--lazy-logging"])):
        out = str(tmp_path / name)
        train.main(["--run-name", name, "--n-epochs", "6", "--cuda", "--track", TRACKS["big_track"], "--n-envs", "256", "--n-steps", "64", "--batch-size", "64",
                    "--train-iters", "2", "--num-rays", "16", "--out-dir", out] + extra)
        _, lg = _run_dirs(out)
        rows[name] = [json.loads(l) for l in open(os.path.join(out, "logs", lg[0], "scalars.jsonl"))]
    assert len(rows["sync"]) == len(rows["lazy"]) == 6
    for a, b in zip(rows["sync"], rows["lazy"]):
        for k in ("losses/total_loss", "charts/avg_reward", "charts/learning_rate", "global_step"):
            assert a[k] == pytest.approx(b[k], rel=1e-6, abs=1e-9), k


def test_train_cli_end_to_end_and_resume(tmp_path):
    import train
    out = str(tmp_path)
    base = ["--cuda", "--track", TRACKS["big_track"], "--n-envs", "256", "--n-steps", "64", "--batch-size", "64", "--train-iters", "2",
            "--num-rays", "16", "--out-dir", out]
    train.main(["--run-name", "t1", "--n-epochs", "10"] + base)
    ck, lg = _run_dirs(out)
    assert len(ck) == 1 and ck == lg and ck[0].endswith("_t1")                       # train.py:118-123: <timestamp>_<run-name>
    files = sorted(os.listdir(os.path.join(out, "checkpoints", ck[0])))
    assert files == ["checkpoint_10.dat", "model.dat", "trainer_10.pt"]              # train.py:283,301 (+ the resumable state)
    sd = torch.load(os.path.join(out, "checkpoints", ck[0], "model.dat"), weights_only=True)
    assert set(sd) == {"actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias",
                       "critic.0.weight", "critic.0.bias", "critic.2.weight", "critic.2.bias"}
    assert sd["actor.0.weight"].shape == (256, 23)
    hp = open(os.path.join(out, "logs", ck[0], "hyperparameters.md")).read()         # train.py:132-135
    assert hp.startswith("|param|value|\n|-|-|\n") and "|n_envs|256|" in hp and "|learning_rate|0.0003|" in hp
    rows = [json.loads(l) for l in open(os.path.join(out, "logs", ck[0], "scalars.jsonl"))]
    assert len(rows) == 10
    for tag in ("losses/policy_loss", "losses/value_loss", "losses/entropy", "losses/total_loss", "charts/avg_reward",
                "charts/learning_rate", "charts/SPS"):                               # train.py:286-292
        assert all(np.isfinite(r[tag]) for r in rows), tag
    assert rows[-1]["global_step"] == 10 * 256 * 64
    assert rows[-1]["charts/learning_rate"] == pytest.approx(3e-4 * 0.99 ** 10, rel=1e-5)
    # --resume through the CLI: 2 more epochs continue the run (epoch counter, step counter, lr schedule)
    resume = os.path.join(out, "checkpoints", ck[0], "trainer_10.pt")
    train.main(["--run-name", "t2", "--n-epochs", "12", "--resume", resume] + base)
    ck2, _ = _run_dirs(out)
    new = [d for d in ck2 if d.endswith("_t2")]
    assert len(new) == 1
    rows2 = [json.loads(l) for l in open(os.path.join(out, "logs", new[0], "scalars.jsonl"))]
    assert len(rows2) == 2 and rows2[-1]["global_step"] == 12 * 256 * 64
    assert rows2[-1]["charts/learning_rate"] == pytest.approx(3e-4 * 0.99 ** 12, rel=1e-5)
    # ... and equals an uninterrupted 12-epoch run, bit for bit
    train.main(["--run-name", "t3", "--n-epochs", "12"] + base)
    ck3, _ = _run_dirs(out)
    full = [d for d in ck3 if d.endswith("_t3")][0]
    a = torch.load(os.path.join(out, "checkpoints", new[0], "model.dat"), weights_only=True)
    b = torch.load(os.path.join(out, "checkpoints", full, "model.dat"), weights_only=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("kw", [dict(), dict(custom_mlp=False), dict(fused_update=False), dict(use_graphs=False),
                                dict(custom_mlp=False, use_graphs=False)],
                         ids=["custom-epoch-graph", "torch-mlp-graphs", "torch-update-graphs", "custom-eager", "torch-mlp-eager"])
def test_checkpoint_resume_on_every_update_path(tmp_path, kw):
    """3 epochs in one go == 2 epochs, save, fresh trainer, load, 1 more -- also on the paths that build their graphs lazily
    (torch MLPs between the fused kernels, capturable torch Adam): the warm-up of the graph capture must not disturb a
    restored optimizer state."""
    cfg = PPOConfig(n_envs=256, n_steps=48, batch_size=64, train_iters=3, track=TRACKS["big_track"], num_rays=16, seed=5, **kw)
    a = Trainer(cfg, device="cuda")
    for _ in range(3):
        a.run_epoch(sync=False)
    b = Trainer(cfg, device="cuda")
    for _ in range(2):
        b.run_epoch(sync=False)
    torch.cuda.synchronize()
    torch.save(b.state_dict(), tmp_path / "t.pt")
    b.close()
    c = Trainer(cfg, device="cuda")
    c.load_state_dict(torch.load(tmp_path / "t.pt", map_location="cuda", weights_only=False))
    c.run_epoch(sync=False)
    torch.cuda.synchronize()
    exact = kw.get("fused_update", True)      # torch's capturable Adam recomputes its bias correction on the device: 1e-6 slack
    if exact:
        assert torch.equal(a.learner.flat_param, c.learner.flat_param)
    else:
        assert torch.allclose(a.learner.flat_param, c.learner.flat_param, atol=2e-6, rtol=1e-5)
    assert torch.equal(a.buffer.act_buf, c.buffer.act_buf) if exact else True
    a.close(); c.close()


# ------------------------------------------------------------------------------------------------
# gymnasium-shaped surface
# ------------------------------------------------------------------------------------------------
def test_spaces_and_info_keys():
    env = pc.VecCarEnv(64, TRACKS["big_track"], num_rays=16, reward_scaling=0.1)
    assert env.single_observation_space.shape == (23,) and env.single_action_space.n == 9      # train.py:141-142
    assert env.observation_space.shape == (64, 23) and env.num_envs == 64
    # the Box bounds CarEnv.__init__ declares (car_env.py:514-524): low = [0, 0, -1, -1, -1, -1, 0 ...], high = 1, float32; the
    # reference's own declared width is 6 + num_rays = 22 while it produces 23 entries (quirk Q1)
    sp = env.single_observation_space
    assert sp.low.dtype == np.float32 and sp.high.dtype == np.float32 and sp.low.shape == (23,) == sp.high.shape
    assert sp.low.tolist() == [0.0, 0.0, -1.0, -1.0, -1.0, -1.0] + [0.0] * 17 and sp.high.tolist() == [1.0] * 23
    assert sp.declared_shape == (22,) and env.observation_space.low.shape == (64, 23)
    assert env.single_action_space.contains(np.int64(8)) and not env.single_action_space.contains(np.int64(9))
    obs, info = env.reset()
    assert info == {} and obs.shape == (64, 23)
    assert sp.contains(obs[0].cpu().numpy()) and env.observation_space.contains(obs.cpu().numpy())
    ora = oracle.OracleVecEnv(oracle.Track(TRACKS["big_track"]), 64, num_rays=16, reward_scaling=0.1)
    ora.reset()
    rng = np.random.default_rng(0)
    saw_reset = False
    for t in range(120):
        a = rng.choice([0, 4, 5, 2], size=64).astype(np.int64)
        _, _, term, trunc, info = env.step(torch.from_numpy(a).cuda(), info=True)
        _, _, TE, TR = ora.step(a)
        # CarEnv._get_info() (car_env.py:599-603) of every env's state after the step (auto-reset envs: 0 / 0)
        assert np.array_equal(info["time_passed"].cpu().numpy(), ora.time_step)
        assert np.array_equal(info["gates_passed"].cpu().numpy(), ora.passed)
        done = (term.cpu().numpy() != 0) | (trunc.cpu().numpy() != 0)
        assert np.array_equal(done, TE | TR)
        if done.any():
            saw_reset = True
            assert np.all(info["time_passed"].cpu().numpy()[done] == 0)
    assert saw_reset
    i2 = env.infos()
    assert np.array_equal(i2["time_passed"].cpu().numpy(), ora.time_step)
    env.close()


# ------------------------------------------------------------------------------------------------
# multi-rank update: the all-reduce inside the epoch graph
# ------------------------------------------------------------------------------------------------
def _one_rank_collective_worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    for name, kw in (("captured", dict(capture_collectives=True)), ("eager", dict(capture_collectives=False))):
        cfg = PPOConfig(n_envs=512, n_steps=64, batch_size=32, train_iters=3, track=TRACKS["big_track"], num_rays=16, seed=11,
                        force_collective=True, **kw)
        tr = Trainer(cfg, device="cuda:0")
        for _ in range(3):
            tr.run_epoch(sync=False)
        torch.cuda.synchronize()
        res[name] = (tr.learner.flat_param.cpu(), tr.learner.metrics.cpu(), tr.learner._epoch_graph is not None,
                     tr.learner._capture_failed)
        tr.close()
    cfg = PPOConfig(n_envs=512, n_steps=64, batch_size=32, train_iters=3, track=TRACKS["big_track"], num_rays=16, seed=11)
    tr = Trainer(cfg, device="cuda:0")
    for _ in range(3):
        tr.run_epoch(sync=False)
    torch.cuda.synchronize()
    res["single"] = (tr.learner.flat_param.cpu(),)
    tr.close()
    torch.save(res, os.path.join(out_dir, "res.pt"))
    dist.destroy_process_group()


def test_gradient_all_reduce_is_captured_into_the_update_graph(tmp_path):
    """The multi-rank update path (K10, K11, RCCL all-reduce, clip + Adam per minibatch) on a ONE-rank RCCL communicator:
    with capture_collectives the whole epoch's update -- collectives included -- is one HIP graph replay; it must give the
    bits of the eagerly enqueued sequence, and agree with the single-rank kernels (different norm summation: 1e-6)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_one_rank_collective_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / "res.pt")
    assert res["captured"][2] and not res["captured"][3], "the RCCL all-reduce was not captured into the epoch graph"
    assert not res["eager"][2]
    assert torch.equal(res["captured"][0], res["eager"][0]) and torch.equal(res["captured"][1], res["eager"][1])
    assert torch.allclose(res["captured"][0], res["single"][0], atol=2e-6, rtol=1e-5)


def _two_gpu_worker(rank, world, port, out_dir, exchange, capture):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    cfg = PPOConfig(n_envs=512, n_steps=64, batch_size=32, train_iters=2, track=TRACKS["big_track"], num_rays=16, seed=11,
                    exchange=exchange, capture_collectives=capture)
    tr = Trainer(cfg, device=f"cuda:{rank}", rank=rank, world_size=world)
    tr.run_epoch()
    s2 = tr.run_epoch()
    torch.cuda.synchronize()
    torch.save({"param": tr.learner.flat_param.cpu(), "acts": tr.buffer.act_buf.cpu(), "scalars": s2,
                "captured": tr.learner._epoch_graph is not None}, os.path.join(out_dir, f"r{rank}_{exchange}_{int(capture)}.pt"))
    tr.close()
    dist.destroy_process_group()


# What the two tests below verify ON A BOX WITH TWO GPUs (this pool's boxes have one; the driver's 8-GPU node runs bench.py, not
# pytest): SURVEY 8(e)'s one exchange step per minibatch ACROSS DEVICES -- one process per GPU, RCCL communicator of world size 2
# over xGMI, or the library's one-shot exchange writing into the peer GPU's hipIpc-mapped staging buffer (system-scope release /
# acquire across the fabric, peer access enabled by pc_xchg_connect) -- eager and captured into the epoch's update graph: replicas
# bit-identical after two epochs, different action streams per shard, all-reduced scalars equal, and p2p == RCCL bit for bit (two
# ranks: a + b in either order).  Same-device rehearsals of all of this run in tests/test_trainer_gpu.py.
_TWO_GPU_REASON = ("needs two GPUs (found %d): would run one rank per GPU and check that replicas stay bit-identical through %s "
                   "across devices (xGMI); the same-device rehearsal is test_trainer_gpu.py::%s")


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason=_TWO_GPU_REASON % (torch.cuda.device_count(), "an RCCL all-reduce of world size 2, eager and captured",
                                              "test_two_ranks_on_one_gpu_keep_replicas_identical"))
@pytest.mark.parametrize("capture", [False, True])
def test_two_rccl_ranks_keep_replicas_identical(tmp_path, capture):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_two_gpu_worker, args=(2, port, str(tmp_path), "rccl", capture), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}_rccl_{int(capture)}.pt") for r in (0, 1))
    assert torch.equal(r0["param"], r1["param"]) and not torch.equal(r0["acts"], r1["acts"])
    assert r0["scalars"]["charts/avg_reward"] == pytest.approx(r1["scalars"]["charts/avg_reward"])
    assert r0["captured"] == capture


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason=_TWO_GPU_REASON % (torch.cuda.device_count(), "the one-shot p2p exchange (pc_xchg_*) over peer-mapped buffers, eager and captured",
                                              "test_one_shot_p2p_exchange_between_two_ranks_on_one_gpu"))
@pytest.mark.parametrize("capture", [False, True])
def test_one_shot_p2p_exchange_across_two_gpus(tmp_path, capture):
    import torch.multiprocessing as mp
    res = {}
    for exchange in ("p2p", "rccl"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_two_gpu_worker, args=(2, port, str(tmp_path), exchange, capture and exchange == "p2p"), nprocs=2, join=True)
        res[exchange] = [torch.load(tmp_path / f"r{r}_{exchange}_{int(capture and exchange == 'p2p')}.pt") for r in (0, 1)]
    p0, p1 = res["p2p"]
    assert torch.equal(p0["param"], p1["param"]) and not torch.equal(p0["acts"], p1["acts"])
    assert torch.equal(p0["param"], res["rccl"][0]["param"])              # the very bits of the RCCL path
    assert p0["captured"] == capture


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent spawns the ranks as child processes before touching the GPU.
    On a 1-GPU box the two ranks share cuda:0 over gloo (--same-device); the JSON line must come from rank 0 of the children."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--workload",
                        "cfg1", "--n-steps", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["n_envs_total"] == 8192
    assert d["config"]["backend"] == "gloo" and d["config"]["rccl_ranks"] == 0      # the line names the transport that really ran
    assert d["value"] > 0 and d["scaling"] == "weak"


# ------------------------------------------------------------------------------------------------
# the driver's multi-GPU invocation, rehearsed end to end on one device
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("exchange,capture", [("p2p", True), ("rccl", False)])
def test_bench_self_spawned_two_ranks_on_one_device(exchange, capture):
    """`python bench.py --gpus 2` the way the driver's N > 1 run enters it when no launcher is around: the parent (which never
    touches the GPU) starts `python -m torch.distributed.run` as a CHILD, two ranks come up, train, and rank 0 prints exactly one
    JSON line.  Rehearsal form: both ranks on cuda:0, gloo for the rendezvous; the gradient exchange per minibatch is the
    library's one-shot all-reduce over hipIpc-mapped buffers captured into the epoch graph (p2p), or torch.distributed's
    all_reduce enqueued eagerly (what `nccl` = RCCL takes on real multi-GPU nodes).  Checked: exit code 0, one line, two ranks,
    no RCCL communicator claimed, the exchange named, the replicas bit-identical after the timed epochs (train.py:259-261)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--exchange", exchange,
           "--workload", "cfg1", "--n-envs", "1024", "--n-steps", "64", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
    if capture:
        cmd.append("--capture-collectives")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line on stdout, nothing else (RCCL banners, warnings: stderr)
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["ranks"] == 2 and c["n_envs_total"] == 2048 and out["scaling"] == "weak"
    assert c["rccl_ranks"] == 0 and c["backend"] == "gloo"          # no RCCL communicator exists in this rehearsal, and none is claimed
    assert ("one-shot" in c["gradient_exchange"]) == (exchange == "p2p")
    assert ("captured" in c["update_path"]) == capture and "multi-rank" in c["update_path"]
    assert c["replicas_bit_identical"] is True
    assert out["value"] > 0 and c["rollout"] == "mega"


@pytest.mark.parametrize("world,exchange", [(4, "rccl"), (4, "p2p")])
def test_bench_self_spawned_four_ranks_on_one_device(world, exchange):
    """The driver's N > 1 entry at world size 4, rehearsed on one device (this pool lets one job put six processes on a card, the test
    runner included; the 8-way line differs only in the count): exit code 0, ONE JSON line, the ranks counted, no RCCL communicator claimed over
    gloo, the replicas bit-identical -- and the fields the first real multi-GPU run needs to explain itself: `exchange_us`
    {mean, p50, p90, max} by HIP events around every eagerly enqueued exchange, the rollout's min / max over the ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--same-device", "--backend", "gloo", "--exchange", exchange,
           "--workload", "cfg1", "--n-envs", "1024", "--n-steps", "64", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == world and c["ranks"] == world and c["n_envs_total"] == 1024 * world and c["rccl_ranks"] == 0 and c["backend"] == "gloo"
    assert c["replicas_bit_identical"] is True and ("one-shot" in c["gradient_exchange"]) == (exchange == "p2p")
    ex = c["exchange_us"]
    assert ex["n"] == 2 * 40 * 1 and 0 < ex["p50"] <= ex["p90"] <= ex["max"] and ex["mean"] > 0, ex      # 2 timed epochs x 40 iterations x ceil(64 / 512) minibatches
    ro = c["rollout_ms_over_ranks"]
    assert len(ro["per_rank"]) == world and 0 < ro["min"] <= ro["max"]
    assert c["visible_devices"] >= 1


def test_bench_one_rank_rccl_communicator_reports_live_ranks_and_exchange_latency():
    """`bench.py --force-collective` on ONE rank with backend nccl: the multi-rank update path (K10, K11, RCCL all-reduce, clip + Adam per
    minibatch, eagerly enqueued) over a real RCCL communicator of size one -- the closest a one-GPU box gets to the N > 1 line's RCCL
    fields: `rccl_ranks` counted by an all-reduce of ones on the LIVE communicator (1), `exchange_us` from HIP events around every
    all-reduce of the timed epochs (the transport-free floor of the figure the multi-GPU run will report)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--backend", "nccl", "--workload", "cfg1", "--n-envs", "1024",
                        "--n-steps", "64", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    c = json.loads(lines[0])["config"]
    assert c["backend"] == "nccl" and c["rccl_ranks"] == 1 and c["ranks"] == 1 and "multi-rank" in c["update_path"]
    ex = c["exchange_us"]
    assert ex["n"] == 80 and 0 < ex["p50"] <= ex["max"], ex
    print("one-rank RCCL all-reduce of the 59 KB bucket, HIP events:", {k: round(v, 1) for k, v in ex.items() if isinstance(v, float)})


def test_bench_refuses_more_gpus_than_are_visible():
    """`--gpus N` with fewer than N devices visible and no --same-device: a one-line reason on stderr and a non-zero exit code from the
    PARENT, before any rank is started (the first real multi-GPU run must not die somewhere inside a rendezvous for this reason)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(2, n))], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "device(s) are visible" in r.stderr and "--same-device" in r.stderr


def test_bench_default_line_carries_every_baseline_config_and_the_exact_dtype():
    """`python bench.py` as the driver runs it (shortened: 2 timed epochs): ONE JSON line whose headline is the target workload and
    which also holds the bit-exact dtype's value on the same workload, every other single-GPU BASELINE config with its own
    ms_per_step / launch duration / roofline fraction, the all-usable-cores CPU baseline with its core counts, and the live
    parity check -- measured, finite, from the kernels the headline claims (persistent launches; for f64 K9's literal form, at
    configs[1] inside the small form), and every other config in the bit-exact dtype as well."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "2"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("env steps/sec") and d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["rollout"] == "mega" and d["value"] > 1e9
    rf = d["roofline"]
    assert rf["bound"] == "valu" and 0.1 < rf["frac"] < 1.0 and rf["launch_us"] > 0 and rf["traffic"] is not None
    # the env alone (SURVEY 8(d) level (i)), on the trainer's env state and the last rollout's actions: pc_env_step as the boundary launches
    # it (the table-driven kernel at this batch size), the generic kernel beside it, and T steps in one launch
    k1, eo = rf["k1_standalone"], rf["env_only"]
    assert "K1f" in k1["kernel"] and 0 < k1["launch_us"] < k1["generic_kernel_launch_us"] and "trainer's env state" in k1["state"], k1
    assert "K1f" in eo["kernel"] and 0 < eo["us_per_step"] < k1["launch_us"] and eo["value"] > d["value"] and 0 < eo["valu_frac"] < 1, eo
    assert d["parity_check"]["ok"] is True
    rb = d["parity_check"]["rare_branches"]      # the second leg: injected states + the trained policy, the same kernel
    assert rb["ok"] is True and rb["events_replayed"]["laps"] > 0 and rb["events_replayed"]["truncations"] > 0 and rb["events_replayed"]["terminated_at_time_limit"] > 0, rb
    e = d["exact_f64_value"]
    assert e["dtype"] == "f64" and e["rollout"] == "mega" and e["kernel"] == "K9-literal" and e["value"] > 1e9 and 0 < e["roofline"]["frac"] < rf["frac"]
    assert d["config"]["rollout_kernel"] == "K9"
    ow = d["other_workloads"]
    assert set(ow) == {"cfg1", "cfg2", "cfg4", "cfg4i"}
    for k, v in ow.items():
        assert "error" not in v, (k, v)
        assert v["rollout"] == "mega" and v["value"] > 1e8 and v["ms_per_step"] > 0 and 0 < v["roofline"]["frac"] < 1 and v["roofline"]["flops_per_env_step"] > 0
        x = v["exact_f64"]
        assert "error" not in x, (k, x)
        assert x["kernel"] == {"cfg1": "K9s-literal", "cfg2": "K9-literal", "cfg4": "K9m-literal", "cfg4i": "K9m-literal"}[k] and v["kernel"] == {"cfg1": "K9s", "cfg2": "K9", "cfg4": "K9m", "cfg4i": "K9m"}[k] and x["value"] > 0.2 * v["value"]      # (short side measurements on a box that has just started: either can be off by 30 %)
    assert "33 actual" in ow["cfg2"]["workload"] and "track.json + big_track.json (halves)" in ow["cfg4"]["workload"] and "interleaved" in ow["cfg4i"]["workload"]
    assert 0.1 < ow["cfg4i"]["ratio_to_halves_layout"] < 1.2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["threads"] == c["cores"] == c["usable_cores"] <= c["host_cores"]
    assert c["env_only_value"] > 0 and c["env_only_one_thread_value"] > 0 and c["value"] > 0      # (how they compare is the host's business)
    assert d["strict_fp32_value"]["value"] > 0 and d["fp32_grade_bf16x3_value"]["value"] > 0
    sf = d["strict_f64_fp32_value"]      # float64 env + exact fp32 policy chain: one persistent launch as well
    assert sf["dtype"] == "f64" and sf["rollout"] == "mega" and sf["kernel"] == "K9-literal" and sf["value"] > 5e8, sf
