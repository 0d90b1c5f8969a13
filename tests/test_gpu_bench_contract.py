"""bench.py's output contract, checked on the line itself: the driver's own invocation (`--gpus 1 --steps 20 --warmup 5`) and
the self-launching multi-rank form (`--gpus 2` from a plain shell; on a 1-GPU box the two ranks share the device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600, **extra_env):
    env = dict(os.environ, **{"RLARM_BENCH_ALTERNATIVES": "0", **extra_env})   # the extra passes have a test of their own
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], cwd=REPO, env=env, capture_output=True,
                       text=True, timeout=timeout)
    if p.returncode != 0:     # the ranks' own messages come first, torchrun's failure banner (long, generic) last
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "bench_contract_failure.log"), "w") as fh:
            fh.write(p.stdout + "\n==== stderr ====\n" + p.stderr)
        own = [ln for ln in p.stderr.splitlines() if "elastic" not in ln and not ln.startswith(("E ", " "))]
        raise AssertionError("\n".join(own[-60:]) + "\n....\n" + p.stderr[-1500:])
    # (the ranks' constructor prints -- the reference's own "Buffer_size: ..." line -- may interleave; the record is the one
    # line that parses as the contract's JSON object)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line
    assert p.stdout.rstrip().splitlines()[-1] == lines[0]   # ... and it is the last thing printed
    return json.loads(lines[0])


def test_driver_invocation_prints_the_contract_line():
    with open(os.path.join(REPO, "BASELINE.json")) as fh:
        base = json.load(fh)
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "3")
    assert d["metric"] == base["metric"] and d["unit"] == "transitions/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 256 * 20 / (d["ms_per_step"] * 20e-3)) <= 1e-3 * d["value"]     # value = transitions / timed region
    assert 2e6 < d["value"] < 2e7                                     # a plausible MI355X figure, not a unit slip
    eng = d["config"]["engine"]
    assert (eng["engine"], eng["slab_rows"], eng["weight_grad"]) == ("slab8", 4, "gemm_lds 32x32")
    assert eng["launches_per_update"].startswith("split")          # round 4: the default form at the headline shape
    assert d["roofline"]["kernel"] == "k_fb_split8<0>" and eng["kernels_per_update"] == ["k_fb_split8<0>", "k_gemm_lds_adam"]
    cal = d["calibration"]
    assert 0.5 < cal["launch_floor_us"] < 10 and 20 < cal["lds_dma_GBps_per_cu"] < 400 and 4 < cal["mfma4x4_dependent_cycles"] < 40
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == pytest.approx(157.3)
    assert 0.0 < r["frac"] < 1.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert r["achieved"] == pytest.approx(r["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12, rel=1e-3)
    assert r["traffic"] is None or r["traffic"] > 0
    assert 0.0 < r["whole_update_frac"] < r["frac"] * 1.5
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "transitions/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # round 5: the 20 timed steps are the LAST 20 updates of a cycle -- its soft target update is inside the region, and the line
    # says which of its two durations decided the roofline fraction
    assert d["cycle_boundaries_in_timed_region"]["polyak"] == 1 and "placement" in d["cycle_boundaries_in_timed_region"]
    assert r["duration_source"] in ("live", "committed") and r["duration_stale"] in (True, False)
    assert c["config1"]["cores"] == 1 and c["config1"]["value"] > 0 and "100-episode" in c["config1"]["sample"]
    f = d["roofline_sample_kernel_fused"]
    assert f["bound"] == "hbm" and f["bytes_per_transition"] == 528 and 0.0 < f["frac"] < 1.0
    assert f["achieved"] == pytest.approx(528 * f["batch"] / (f["avg_launch_us"] * 1e-6) / 1e9, rel=1e-3)


@pytest.mark.parametrize("batch", [256, 3072])     # 3072: the 32-row engine + split weight-gradient tiles under the exchange
def test_plain_shell_multi_rank_launch(batch):
    d = _run("--gpus", "2", "--batch", str(batch), "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile")
    assert d["n_gpus"] == 2 and d["steps"] == 80
    assert d["config"]["exchange"] in ("peer-memory", "rccl", "torch.distributed")
    assert abs(d["value"] - 2 * batch * 80 / (d["ms_per_step"] * 80e-3)) <= 1e-3 * d["value"]   # whole-job aggregate
    if batch == 3072:     # both forms of the peer exchange end in the same bits
        e = _run("--gpus", "2", "--batch", str(batch), "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile",
                 RLARM_PEER_PHASES="2")
        if d["config"]["exchange"] == "peer-memory":
            assert e["config"]["peer_exchange_form"].startswith("two-phase")
            assert e["config"]["final_losses"] == d["config"]["final_losses"]


@pytest.mark.parametrize("world", [4, 8])
def test_plain_shell_launch_rehearses_4_and_8_ranks(world):
    """BASELINE configs 4 / 5 run on 8 GPUs, which only the driver has: this rehearses the exact code path `bench.py --gpus 4|8`
    takes there -- self-launch under torch.distributed.run, one process per rank, seed + rank sampler streams, parameter
    broadcast, per-update gradient exchange, per-cycle normalizer exchange -- with every rank on the one device of the test box
    (IPC-mapped peer memory exactly as between GPUs; the fabric itself is what cannot be rehearsed).  Nothing is forced, so the
    transport and form are the DEFAULT ones: peer memory, two-phase (reduce-scatter + all-gather) from 4 ranks.  Then the
    one-shot form and the torch.distributed fallback: same replicas, and one-shot == two-phase bit for bit."""
    common = ("--gpus", str(world), "--episodes", "64", "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile")
    d = _run(*common, timeout=900)
    assert d["n_gpus"] == world and d["config"]["global_batch"] == 256 * world
    assert d["config"]["exchange"] == "peer-memory", d["config"]
    assert d["config"]["peer_exchange_form"].startswith("two-phase")
    assert d["config"]["cycle_mode"] == "hipGraph"
    assert d["config"]["replicas_bit_identical"] is True and d["config"]["devices_shared_by_ranks"] is True
    assert abs(d["value"] - world * 256 * 80 / (d["ms_per_step"] * 80e-3)) <= 1e-3 * d["value"]
    assert all(abs(x) < 1e3 for x in d["config"]["final_losses"])
    one = _run(*common, timeout=900, RLARM_PEER_PHASES="1")
    assert one["config"]["peer_exchange_form"].startswith("one-shot") and one["config"]["replicas_bit_identical"] is True
    assert one["config"]["final_losses"] == d["config"]["final_losses"]          # same rank-ordered float32 sums
    t = _run(*common, timeout=900, RLARM_COMM="torch")
    assert t["config"]["exchange"] == "torch.distributed" and t["config"]["replicas_bit_identical"] is True
    for got, want in zip(t["config"]["final_losses"], d["config"]["final_losses"]):  # another summation order across ranks
        assert abs(got - want) <= 2e-2 * max(abs(want), 1e-2)


@pytest.mark.parametrize("name,flags,per_rank_batch", [
    ("config4", ("--batch", "1024"), 1024),
    ("config5", ("--batch", "512", "--replay-k", "8", "--feeder-episodes", "8"), 512)])
def test_baseline_configs_4_and_5_rehearsed_at_their_real_per_rank_workloads(name, flags, per_rank_batch):
    """VERDICT r03 item 1a.  BASELINE config 4 = 8 ranks x (batch 1024, 5000-episode shard), config 5 = 8 ranks x (batch 512,
    replay_k 8, 8 fresh episodes per cycle from the host feeder, 5000-episode shard) -- so far eight ranks had only run at batch
    256 on 64-episode shards.  Here: the exact `bench.py --gpus 8` code path at those per-rank workloads with NOTHING forced, all
    ranks on the one device of the test box (default transport = peer memory with gate kernels, default form at 8 ranks =
    two-phase, cycle = hipGraph), replicas bit-identical, no fallback; then the one-shot form: the same losses bit for bit."""
    common = ("--gpus", "8", *flags, "--steps", "40", "--warmup", "40", "--no-cpu-baseline", "--no-profile")
    d = _run(*common, timeout=1500)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["global_batch"] == 8 * per_rank_batch
    assert c["exchange"] == "peer-memory", c                      # no fallback to RCCL / torch.distributed
    assert c["peer_exchange_form"].startswith("two-phase") and c["cycle_mode"] == "hipGraph"
    assert c["replicas_bit_identical"] is True and c["devices_shared_by_ranks"] is True and c["peer_gate_kernels"] is True
    assert abs(d["value"] - 8 * per_rank_batch * 40 / (d["ms_per_step"] * 40e-3)) <= 1e-3 * d["value"]
    assert all(abs(x) < 1e3 for x in c["final_losses"])
    one = _run(*common, timeout=1500, RLARM_PEER_PHASES="1")
    assert one["config"]["exchange"] == "peer-memory" and one["config"]["peer_exchange_form"].startswith("one-shot")
    assert one["config"]["replicas_bit_identical"] is True
    assert one["config"]["final_losses"] == c["final_losses"]     # one-shot == two-phase: the same rank-ordered float32 sums


def test_multi_rank_launch_survives_a_failing_exchange_on_one_rank():
    """If the first transport fails on ANY rank in the first (untimed) cycle, all ranks agree and rebuild on the next one down
    instead of leaving the scaling run without a line."""
    d = _run("--gpus", "2", "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile", RLARM_BENCH_FAIL_FIRST="1")
    assert d["n_gpus"] == 2 and d["config"]["exchange"] in ("rccl", "torch.distributed")


def test_eight_rank_rehearsal_times_the_exchange_alternatives():
    """there may be exactly one multi-GPU run, so after the timed region of the transport that won, the same
    process group times a short pass on every OTHER exchange (RCCL in the cycle graph, the peer exchange's other forms,
    torch.distributed) and records us/update + replica identity for each; the per-rank PCI bus ids and the rank counts say
    whether it was N ranks on N devices.  Rehearsed with eight ranks on the one device: RCCL cannot run there (it refuses two
    ranks per device; the record must say what the pass ran as), every peer form and the torch fallback must."""
    d = _run("--gpus", "8", "--episodes", "64", "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile", timeout=1500,
             RLARM_BENCH_ALTERNATIVES="1")
    c = d["config"]
    assert d["n_gpus"] == 8 and c["replicas_bit_identical"] is True
    dev = c["devices"]
    assert len(dev["pci_bus_ids_by_rank"]) == 8 and dev["distinct_devices"] == 1 and dev["process_group_ranks"] == 8
    alts = d["exchange_alternatives"]
    assert len(alts) == 5 and all("exchange" in x for x in alts)
    done = [x for x in alts if "us_per_update" in x]
    assert len(done) >= 4, alts                                   # the peer forms + torch (+ whatever the rccl pass fell back to)
    for x in done:
        assert x["replicas_bit_identical"] is True and 10 < x["us_per_update"] < 1e5 and x["ran_as"]
    forms = {x["ran_as"].split(",")[0] + (x["ran_as"].split("phases")[1][:2] if "phases" in x["ran_as"] else "") for x in done}
    assert {"peer-memory 1", "peer-memory 2", "torch.distributed"} <= forms, forms


def test_alternatives_pass_survives_a_failing_transport():
    """... and a transport that fails in the extra passes costs its own entry, not the line."""
    d = _run("--gpus", "2", "--episodes", "64", "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile", timeout=900,
             RLARM_BENCH_ALTERNATIVES="1", RLARM_BENCH_FAIL_ALT="peer memory, two-phase", RLARM_PEER_TIMEOUT_S="3")
    alts = d["exchange_alternatives"]
    bad = [x for x in alts if x["exchange"].startswith("peer memory, two-phase")]
    assert len(bad) == 1 and "error" in bad[0] and "injected" in bad[0]["error"]
    assert sum("us_per_update" in x for x in alts) >= 3


def _trace_kernel_names(flags, env, tmp_path):
    """Kernel names rocprofv3 --kernel-trace lists for `bench.py flags` (short form, as tools/trace_summary.py prints them)."""
    import shutil
    import sqlite3
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    out = str(tmp_path / "trace")
    e = dict(os.environ, TMPDIR="/tmp", RLARM_BENCH_ALTERNATIVES="0", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", out, "-o", "t", "--", sys.executable,
                        os.path.join(REPO, "bench.py"), *flags], cwd="/tmp", env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, p.stdout[-1500:]
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
    assert dbs, os.listdir(out)
    names = set()
    for (nm,) in sqlite3.connect(dbs[0]).execute("select distinct name from kernels"):
        n = nm[5:] if nm.startswith("void ") else nm
        names.add(n.split("(")[0].split("::")[-1])
    return json.loads(line[0]), names


@pytest.mark.parametrize("transport,chain,tiles", [("peer", "k_fb_split8<1>", "k_gemm_lds_adam_peer"), ("rccl", "k_fb_split8<2>", "k_gemm_lds")])
def test_forced_data_parallel_line_names_the_kernels_it_ran(transport, chain, tiles, tmp_path):
    """VERDICT r05 items 1 + 2.  A data-parallel rank takes the split launch (round 6) and the line says so with the kernels the
    library's launch logic enqueues WHILE the transport is attached (round 5 asked after close_comm() and named the single-rank
    k_fb_split8 for a rank that ran k_fb_slab8): `config.engine.kernels_per_update` = hp_agent_update_kernels, the roofline's
    kernel is one of them, and a rocprofv3 --kernel-trace of the same command lists every one of them."""
    flags = ("--steps", "400", "--warmup", "80", "--no-cpu-baseline")
    d, traced = _trace_kernel_names(flags, {"RLARM_BENCH_FORCE_DP": "1", "RLARM_COMM": transport}, tmp_path)
    c, r = d["config"], d["roofline"]
    assert c["exchange"] == {"peer": "peer-memory", "rccl": "rccl"}[transport]
    kp = c["engine"]["kernels_per_update"]
    assert kp[0] == chain and tiles in kp, kp
    assert c["engine"]["launches_per_update"].startswith("split")
    assert r["kernel"] in kp and r["kernels_per_update"] == kp
    own = [k for k in kp if not k.startswith("rccl:")]             # (RCCL's kernel carries RCCL's own name in the trace)
    assert set(own) <= traced, (own, sorted(traced))
    assert "k_fb_split8<0>" not in traced, sorted(traced)          # nothing of the single-rank form ran (the prologue included)
    assert "k_fb_slab8" not in traced, sorted(traced)              # ... nor of the two-launch form
    assert r["all_matrix_kernels"]["chain"]["kernel"] == chain and 0 < r["frac"] < 1
    assert r["duration_source"] in ("live", "committed")


@pytest.mark.parametrize("inject", ["selfcheck", "ipc"])
def test_first_contact_failures_of_the_peer_exchange_still_yield_a_line(inject):
    """VERDICT r05 item 6.  What only a multi-GPU node can fail for real, forced on the one device: the attach-time self-check
    reporting mismatches (RLARM_PEER_INJECT=selfcheck) and hipIpcOpenMemHandle refusing a peer's handle (=ipc), on one rank.
    Every rank must drop the peer exchange together, the run must degrade to the next transport (RCCL on a device each; two
    ranks on ONE device cannot use RCCL, so torch.distributed here), print the headline with the failure recorded and the
    alternatives pass inside its time box, and exit 0."""
    d = _run("--gpus", "2", "--episodes", "64", "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile", timeout=600,
             RLARM_BENCH_ALTERNATIVES="1", RLARM_BENCH_ALT_BUDGET_S="45", RLARM_PEER_INJECT=inject + "@1")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["exchange"] in ("rccl", "torch.distributed") and c["replicas_bit_identical"] is True
    fb = c["exchange_fallbacks"]
    assert fb and fb[0]["exchange"] == "peer-memory" and ("self-check" if inject == "selfcheck" else "mapping") in fb[0]["refused_at_attach"]
    alts = d["exchange_alternatives"]
    assert len(alts) == 5 and all("exchange" in x for x in alts)
    assert all(("us_per_update" in x) or ("skipped" in x) or ("error" in x) for x in alts)
    assert not any(x.get("ran_as", "").startswith("peer-memory") for x in alts)      # the injection refuses it every time


def test_alternatives_pass_is_time_boxed():
    """... and the extra passes fit their time box: with 12 s for five transports every pass that starts gets a cut number of
    cycles or is recorded as skipped, and the line is printed."""
    d = _run("--gpus", "2", "--episodes", "64", "--steps", "80", "--warmup", "40", "--no-cpu-baseline", "--no-profile", timeout=600,
             RLARM_BENCH_ALTERNATIVES="1", RLARM_BENCH_ALT_BUDGET_S="12")
    alts = d["exchange_alternatives"]
    assert len(alts) == 5 and any("skipped" in x for x in alts), alts
    for x in alts:
        if "us_per_update" in x:
            assert x["steps"] <= 400 and x["replicas_bit_identical"] is True
