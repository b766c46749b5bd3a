"""GPU: bench.py's world > 1 branch (process group, barriers, the MAX / SUM all-reduces, the per-rank gather) executed on a
one-GPU box: MQ_BENCH_FAKE_RANKS=1 puts every rank on device 0 with the gloo backend (CPU tensors for the collectives).  The
launcher is a fresh child process (torch.distributed.run), never a process that has touched the GPU.  Also: mq_index_clone onto
the SAME device at bench size, both replicas mapping at once."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(world, mode, dump=None, reads=6000):
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--genome-scale", "0.02", "--reads", str(reads), "--scaling", mode,
            "--no-cpu-baseline", "--no-e2e", "--no-configs"]
    env = dict(os.environ, MQ_BENCH_FAKE_RANKS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if dump:
        env["MQ_BENCH_DUMP_HITS"] = str(dump)
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_line_two_ranks(mode, tmp_path):
    j = _bench(2, mode, dump=tmp_path)
    assert j["n_gpus"] == 2 and j["scaling"] == mode and j["value"] > 0 and j["overflow_reads"] == 0
    pr = j["per_rank_gbases_s"]
    assert len(pr) == 2 and all(x > 0 for x in pr)
    # value = all ranks' bases over the SLOWEST rank's time: never above the sum of the ranks' own rates, and close to it when the
    # ranks finish together (two ranks time-sharing one device do)
    assert j["value"] <= sum(pr) * 1.001
    assert j["value"] >= 0.5 * sum(pr), (j["value"], pr)
    assert j["roofline"]["bound"] == "hbm"
    h = [np.load(tmp_path / ("hits_rank%d_of_2.npy" % r)) for r in range(2)]
    if mode == "strong":
        # the two shards of the one read set, rank order = read order: together they are what ONE rank maps from the same read set
        j1 = _bench(1, mode, dump=tmp_path)
        assert j1["n_gpus"] == 1
        whole = np.load(tmp_path / "hits_rank0_of_1.npy")
        assert np.array_equal(np.concatenate(h), whole)
        assert j["config"]["reads_per_step_per_gpu"] * 2 >= 6000 - 1
    else:
        assert h[0].shape == h[1].shape and not np.array_equal(h[0], h[1])  # every rank its own batch (seed + rank)


def test_bench_gpus_2_typed_as_the_driver_types_it():
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE: bench.py starts its own ranks (a fresh torch.distributed.run
    child, before anything has touched the GPU) and rank 0's line is the only JSON line; the N = 1 line through the same entry is
    unchanged in shape."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MQ_BENCH_FAKE_RANKS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--genome-scale", "0.02", "--reads", "6000", "--no-cpu-baseline", "--no-e2e", "--no-configs"]
    out = {}
    for n in (2, 1):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n)] + common, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        assert len(lines) == 1, r.stdout[-2000:]
        out[n] = json.loads(lines[0])
    assert out[2]["n_gpus"] == 2 and out[1]["n_gpus"] == 1
    assert out[2]["scaling"] == "weak" and len(out[2]["per_rank_gbases_s"]) == 2
    for key in ("metric", "unit", "dtype", "data", "higher_is_better"):
        assert out[1][key] == out[2][key]
    assert set(out[2]) - set(out[1]) <= {"per_rank_gbases_s"}, set(out[2]) ^ set(out[1])
    assert out[2]["collectives"].startswith("gloo") and out[1]["collectives"] is None


def test_bench_rccl_collectives_on_the_one_gpu():
    """The collectives of the N > 1 path with the backend a real multi-GPU run uses -- "nccl" = RCCL, device tensors -- on the one GPU a test
    box has (MQ_BENCH_FORCE_DIST=1: a one-rank process group): init_process_group, barrier, the MAX and SUM all-reduces and the all-gather
    execute and the line is the N = 1 line (value = the rank's own rate)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "MQ_BENCH_FAKE_RANKS")}
    env.update(MQ_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--genome-scale", "0.02", "--reads", "20000", "--no-cpu-baseline", "--no-e2e",
                        "--no-configs"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["collectives"].startswith("nccl") and len(j["per_rank_gbases_s"]) == 1
    assert abs(j["per_rank_gbases_s"][0] - j["value"]) <= 0.02 * j["value"] and j["records_written"] == 20000


def test_bench_gpus_8_on_one_device_is_affordable():
    """`python bench.py --gpus 8` as the driver's 8-GPU node will run it, here with every rank on the one device (MQ_BENCH_FAKE_RANKS=1) and
    196,608 reads per rank: it has to finish, print ONE line with n_gpus 8, and stay small on the host -- the genome is synthesised once (rank 0,
    into /dev/shm) and mapped by the others, a rank's reads are synthesised slice by slice straight into its device batch and kept nowhere else.
    Bound: 12 GB of peak RSS per rank (the 3.1-GB genome mapping, two 0.9-GB page-locked slice buffers, the runtimes); before round 6 a rank
    held the genome, a 39-GB capacity layout and the 37-GB batch at the default step."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MQ_BENCH_FAKE_RANKS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--reads", "196608", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                       timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["value"] > 0 and j["overflow_reads"] == 0
    assert len(j["per_rank_gbases_s"]) == 8 and len(j["per_rank_setup_s"]) == 8 and len(j["per_rank_peak_rss_gb"]) == 8
    assert max(j["per_rank_peak_rss_gb"]) < 12.0, j["per_rank_peak_rss_gb"]
    assert j["records_written"] == j["config"]["reads_per_step_per_gpu"] == 196608
    assert j["config"]["index_table_bytes"] > 16 * 10**9  # the CHM13-sized index, replicated per rank
    assert j["cpu_baseline"] is None and j["configs"] is None and j["end_to_end"] is None  # rank 0 at N = 1 only
    print("bench.py --gpus 8 on one device: per-rank setup %s s, peak RSS %s GB" % (j["per_rank_setup_s"], j["per_rank_peak_rss_gb"]))


def test_bench_more_ranks_than_gpus_fails_fast():
    """Without the fake-rank hook, --gpus 2 on a one-GPU box must say so at once (every rank, before any work), not hang in RCCL."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "MQ_BENCH_FAKE_RANKS")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--genome-scale", "0.02", "--reads", "1000", "--no-cpu-baseline", "--no-e2e",
                        "--no-configs"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "one rank per GPU" in (r.stdout + r.stderr)


def test_clone_on_the_same_device_at_bench_size(simlib, oracle):
    """mq_index_clone(src, same device) of the CHM13-sized index (a 17 GB table copied inside one device), then the source and the
    replica map the same 32,768-read batch from two threads at once through their own stream slots: identical hits, the oracle's
    columns on a sample."""
    import mapquik_amd as mq
    T = max(2, min(16, len(os.sched_getaffinity(0))))
    g, off, names = simlib.make_genome(list(simlib.CHM13_LIKE), seed=2013, threads=T, repeat_frac=0.05, tandem_frac=0.01)
    P, po = mq.Params(), oracle.params()
    ix = mq.Index(P)
    for r in range(len(names)):
        ix.add_ref(r, names[r], g[int(off[r]):int(off[r + 1])])
    n_unique = ix.finalize()
    assert ix.stats()["table_bytes"] > 16 * 10**9
    rep = ix.clone(0)
    assert rep.stats() == ix.stats()
    reads = simlib.make_reads(g, off, 32768, seed=77, threads=T)
    out, errs = {}, []

    def work(tag, index):
        try:
            ctxs = [index.context() for _ in range(2)]
            res = []
            for k in range(4):
                c = ctxs[k % 2]
                c.submit(reads["bases"], reads["offsets"])
                res.append(c.wait().copy())
            for c in ctxs:
                c.close()
            out[tag] = res
        except Exception as e:  # noqa: BLE001
            errs.append((tag, repr(e)))

    th = [threading.Thread(target=work, args=("src", ix)), threading.Thread(target=work, args=("rep", rep))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    first = out["src"][0].view(np.uint8)
    for tag in ("src", "rep"):
        for h in out[tag]:
            assert np.array_equal(h.view(np.uint8), first)
    ox = oracle.Index()
    ox.build_mt(g, off, names, po, T)
    assert ox.count() == n_unique
    ns = 2048
    offs = reads["offsets"]
    want = ox.map_batch(reads["bases"][:int(offs[ns])], offs[:ns + 1], po, threads=T)
    hits = out["rep"][0][:ns]
    m = want["mapped"] != 0
    assert np.array_equal(hits["status"] == 1, m)
    for a in ("ref_id", "rc", "mapq", "q_start", "q_end", "r_start", "r_end", "score"):
        assert np.array_equal(mq.hit_column(hits, a)[m], want[a][m].astype(np.uint64)), a
    ix.close()
    rep.close()
