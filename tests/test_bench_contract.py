"""-m gpu: bench.py's output contract on the GPU box (one JSON line with the metric of BASELINE.json, `roofline` from live counters with every
fraction <= 1, `cpu_baseline`), and its N > 1 code path exercised end to end with two ranks sharing GPU 0 through gloo (RCCL refuses two ranks on
one device; the 8-GPU run is the driver's)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_line_contract_single_gpu():
    # the census build must come from the kernel sources as they are (build() makes it; a tree where only `make` was run afterwards has a stale one)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_profile
    if not isa_profile.census_available()[0]:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "csrc"), "-s", "census"], timeout=900)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2"], capture_output=True, text=True,
                         cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "value_one_frame_at_a_time"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("C3: 2048x1024") and d["config"]["finite"] is True
    assert abs(d["value"] - 2048 * 1024 / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    assert 100.0 < d["value"] < 1e5 and d["value_one_frame_at_a_time"]["value"] <= d["value"] * 1.15
    # both protocols in the one line; round 6: `value` IS the W + K steps exactly as asked, run first and cold (`value_as_asked` = alias for
    # one round), the same region after the clock pre-warm is `value_prewarmed`
    assert d["value_as_asked"] == d["value"] and d["ms_per_step_as_asked"] == d["ms_per_step"] and d["as_asked"]["value"] == d["value"] and d["protocol"].startswith("as asked")
    assert abs(d["value_prewarmed"] - 2048 * 1024 / d["ms_per_step_prewarmed"] / 1e3) < 1e-6 * d["value_prewarmed"] and d["prewarmed"]["value"] == d["value_prewarmed"]
    assert d["config"]["clock_prewarm_frames"] > 0 and 0.5 * d["value_prewarmed"] < d["value"] < 1.1 * d["value_prewarmed"]
    # the north star's wave early-out as a labelled secondary in every N = 1 line: never the headline, error bounded by eps x radiance
    eo = d["config"]["with_early_out"]
    assert eo["headline"] is False and eo["eps"] == 1e-3 and eo["Mrays_per_s"] > 0.95 * d["value"] and 0.0 < eo["max_abs_err_vs_eps0"] <= 4e-3 and d["config"]["early_out_eps"] == 0.0
    for k in ("value_host_form", "ranks_seen", "per_rank_share_ms"):
        assert k in d, k
    assert d["value_host_form"]["2_in_flight"]["Mrays_per_s"] > d["value_host_form"]["1_in_flight"]["Mrays_per_s"] > 100.0
    r = d["roofline"]
    assert r["pmc"]["collected"], r["pmc"]                                     # the counters were collected in THIS run
    assert r["frac"] is not None and r["timed_region"] is not None, r["census_note"]   # the census ran in THIS run (libcloudsky_census.so: `make census`, built by build())
    # round 4: top level = the dominant kernel ALONE at the guide's issue rates, recomputable in one line from the class totals;
    # frac_calibrated (time-priced, no clock reading) and frac_timed_region (chip-busy over ms_per_step) are named sub-fields
    v = r["valu_issue"]
    assert v["source"] == "census" and v["frame_identical_to_product"] is True
    for unit, ratio in v["census_over_hardware_counters"].items():
        assert ratio is None or abs(ratio - 1.0) < 0.03, (unit, ratio)
    assert v["priced_by_measured_kind_fraction"] > 0.98                        #  nearly every executed VALU instruction is priced by a measurement of its own kind
    assert v["scratch_instructions_per_wavefront"]["plain"] == 0.0             # no scratch access anywhere in clouds_kernel<3,1> (the persistent form: per tile only)
    ka, tr, cl = r["kernel_alone"], r["timed_region"], v["valu_by_class"]
    guide = (2.0 * cl["full"] + 4.0 * cl["half"] + 8.0 * (cl["trans"] + cl.get("quarter", 0)) + 4.0 * cl.get("lane", 0)) / 1024.0
    assert abs(guide - r["achieved"]) < 1e-6 * guide
    assert abs(r["frac"] - guide / (ka["kernel_ms"] * 1e-3 * ka["sclk_mhz"] * 1e6)) < 1e-9 and r["frac"] == ka["frac"]
    assert 0.3 < r["frac"] <= 1.0 and r["frac"] <= r["frac_calibrated"] <= 1.0 and 1500 < ka["sclk_mhz"] < 2600
    assert r["frac"] <= 1.1 * r["frac_timed_region"] and r["frac_timed_region"] <= 1.0 and r["frac_timed_region"] == tr["frac"] and tr["frac"] <= tr["frac_calibrated"] <= 1.05   # (only 6 timed steps here: fill and drain weigh on ms_per_step; the 5 % of slack on the time-priced
    # fraction: its per-kind costs were measured on pure instruction streams, which run the chip at 2.0-2.35 GHz where this kernel runs at 2.38 -- an upper estimate by that ratio, bench.py roofline.note)
    assert 0.5 * d["ms_per_step"] < tr["ms_per_frame_while_sampling"] <= 1.25 * d["ms_per_step"]   # (6 timed steps pay the pipeline's fill and drain; the sampled loop runs >= 0.4 s)
    et = r["executed_tap_bytes"]
    assert et["in_cloud_samples_match_this_run"] and 0.4 < et["over_algorithmic"] < 0.8 and et["bytes_per_launch"] < r["hbm_algorithmic"]["bytes_per_launch_incl_light_march"]
    assert 0.2 < r["l1_gather"]["frac"] <= 1.0 and 0.0 < r["hbm"]["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 1e8
    assert 0.5 < r["kernel_ms_solo"] < 20.0 and r["kernel_ms_in_flight"] >= 0.9 * r["kernel_ms_solo"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


@pytest.mark.parametrize("fif", [4, 8])                      # (the default rotation of two is what test_bench_gpus_2_starts_its_own_ranks runs)
def test_bench_two_ranks_on_one_gpu_through_gloo(fif):
    """fif = 4 / 8: the buffer-set rotation of small rank shares (8 = the default at N = 8, the depth of the rings: ADVICE r4), steps chosen so that
    the drain starts mid-rotation (7 steps: 7 mod 4 = 3, 7 mod 8 = 7)."""
    env = dict(os.environ, CSKY_BENCH_ONE_GPU_DEBUG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4" if fif is None else "7", "--warmup", "1", "--no-cpu-baseline"]
    if fif is not None:
        cmd += ["--frames-in-flight", str(fif)]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["finite"] is True and d["config"]["parallelism"].startswith("bands2")
    assert d["config"]["frames_in_flight"] == (2 if fif is None else fif)
    assert "gathered 2-rank frame vs single-rank frame" in out.stderr             # bench.py compared the gathered frame with a single-context render


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (VERDICT r2 missing 1): bench.py starts its two ranks itself; on this one-GPU box both
    share GPU 0 and gather through gloo (CSKY_BENCH_ONE_GPU_DEBUG=1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CSKY_BENCH_ONE_GPU_DEBUG"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_share_ms"]) == 2 and all(0.05 < x < 50 for x in d["per_rank_share_ms"])
    assert d["gathered_frame_check"]["within_1_fp16_ulp_frac"] >= 0.9999 and d["gathered_frame_check"]["max_abs_diff"] <= 2e-3, d["gathered_frame_check"]
    assert "gathered 2-rank frame vs single-rank frame" in out.stderr


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CSKY_BENCH_ONE_GPU_DEBUG"] = "1"
    return env


def test_bench_gpus_8_process_form():
    """The driver's SCALE command rehearsed at N = 8: `python bench.py --gpus 8 --steps 9 --warmup 2`, self-launched, one process
    per rank, all eight on GPU 0 through gloo.  At N = 8 the defaults are the ones the real run uses: eight frames in flight (a 1/8 share is 4 096
    tiles), the 100 sky-LUT rows split 13/13/13/13/12/12/12/12 behind the bands in the one gather, and 9 timed steps drain mid-rotation (9 mod 8 = 1)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "9", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=ROOT, env=_clean_env(), timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and len(d["per_rank_share_ms"]) == 8 and all(0.05 < x < 50 for x in d["per_rank_share_ms"])
    assert d["config"]["frames_in_flight"] == 8 and d["config"]["parallelism"].startswith("bands8+overlapped-gather") and d["config"]["finite"] is True
    g = d["gathered_frame_check"]
    assert g["within_1_fp16_ulp_frac"] >= 0.9999 and g["max_abs_diff"] <= 2e-3, g
    assert g["sky_lut"] == {"rows_per_rank": 13, "assembled_equals_whole": True}, g
    assert d["value"] == d["value_as_asked"] and d["value_prewarmed"] > 0 and d["steps"] == 9 and d["warmup"] == 2
    assert "gathered 8-rank frame vs single-rank frame" in out.stderr


def test_bench_gpus_8_sweep_in_eight_groups():
    """BASELINE config 5 in the shape the driver would run it at N = 8 with --groups 8: pure frame parallelism over the 64-frame sun sweep, every
    frame gathered on rank 0 (rank 0 keeps 8 x 2 buffer sets in flight)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--groups", "8", "--config", "C5", "--steps", "9", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=ROOT, env=_clean_env(), timeout=1200)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["config"]["frame_groups"] == 8 and d["config"]["finite"] is True
    assert d["gathered_frame_check"]["within_1_fp16_ulp_frac"] >= 0.9999 and d["gathered_frame_check"]["max_abs_diff"] <= 2e-3, d["gathered_frame_check"]


def test_bench_fails_loudly_when_ranks_share_a_device():
    """The real (RCCL) path refuses to time anything when two ranks sit on one device or the communicator is not the size asked for: non-zero return
    code and NO JSON line.  Rehearsed through gloo with the identity check forced on (CSKY_BENCH_FAKE_SHARED_DEVICE=1: both ranks are on GPU 0)."""
    env = _clean_env()
    env["CSKY_BENCH_FAKE_SHARED_DEVICE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")], (out.returncode, out.stdout[-500:])
    assert "distinct devices" in out.stderr
    # ... and a launcher that starts fewer ranks than --gpus says
    env = dict(os.environ, CSKY_BENCH_ONE_GPU_DEBUG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "WORLD_SIZE=2" in out.stderr


@pytest.mark.parametrize("mode", [("8", "1", None), ("4", "2", None), ("3", "1", "--staged"), ("2", "1", None)])
def test_bench_single_process_form(mode):
    """`--single-process`: N contexts behind ONE csky_multi handle (the form a GDExtension host can use), here all on GPU 0.  8 devices x 1 group is
    BASELINE config 4's split; 4 devices in 2 groups is the throughput form (consecutive frames alternate between two 2-way groups); --staged
    uses local band buffers + strided peer copies.  bench.py compares the assembled frame with a single-context render."""
    n, g, extra = mode
    env = dict(os.environ, CSKY_BENCH_ONE_GPU_DEBUG="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n, "--single-process", "--groups", g, "--steps", "9", "--warmup", "2"]
    if extra:
        cmd.append(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == int(n) and d["ranks_seen"] == int(n) and len(d["per_rank_share_ms"]) == int(n)
    assert d["gathered_frame_check"]["within_1_fp16_ulp_frac"] >= 0.9999 and d["gathered_frame_check"]["max_abs_diff"] <= 2e-3, d["gathered_frame_check"]
    assert d["config"]["frame_groups"] == int(g) and d["config"]["finite"] is True and d["config"]["alpha_mean"] > 0.05
    assert "-device frame vs single-context frame" in out.stderr
    ms = d["multi_stats"]                                      # csky_multi_get_stats: preconditions + the per-device timings of one frame
    assert ms["n_devices"] == int(n) and ms["all_peer"] is True and ms["warning"] == "" and ms["peer_access"] == [1] * int(n) and ms["staged"] is (extra == "--staged")
    per = int(n) // int(g)
    timed = [x for x in ms["march_ms"] if x > 0]
    assert len(timed) == per and all(x < 50 for x in timed), ms   # one frame = the devices of ONE group


@pytest.mark.parametrize("cfg", [("4", "2", "C3"), ("3", "3", "C5"), ("2", "2", "C2")])
def test_bench_frame_groups_process_form(cfg):
    """`--groups G` in the one-process-per-GPU form: consecutive frames go to G groups of ranks in turn, each group splits its frame and gathers it
    on rank 0 over its own communicator ({0} + the group).  Self-launched, all ranks on GPU 0 through gloo; G = world is pure frame parallelism
    (BASELINE config 5's sweep: every rank renders whole frames of the 64-frame sun sweep); bench.py checks the LAST frame against a
    single-context render of ITS parameters."""
    n, g, config = cfg
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CSKY_BENCH_ONE_GPU_DEBUG"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n, "--groups", g, "--config", config, "--steps", "7", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    d = _last_json(out.stdout)
    assert d["n_gpus"] == int(n) and d["ranks_seen"] == int(n) and d["config"]["frame_groups"] == int(g) and d["config"]["finite"] is True
    assert d["gathered_frame_check"]["within_1_fp16_ulp_frac"] >= 0.9999 and d["gathered_frame_check"]["max_abs_diff"] <= 2e-3, d["gathered_frame_check"]
    assert "gathered %s-rank frame vs single-rank frame" % n in out.stderr


def test_off_workload_runs_are_flagged_and_carry_no_census():
    """`--shape-noise` / `--coverage` (round 5: what the frame time owes to the stand-in volume, profiles/r05/noise_sensitivity.txt) are NOT the benchmark
    workload: the line says so, its roofline carries no census of another frame, and the in-cloud fraction it reports is the one that explains the time."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-host-form",
                          "--shape-noise", "perlin_freq=8,perlin_octaves=4,dilate=0.8", "--coverage", "0.25"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    c = d["config"]
    assert c["off_workload"] is True and "NOT THE BENCHMARK WORKLOAD" in c["workload"] and c["finite"] is True
    assert d["roofline"]["frac"] is None and d["roofline"]["executed_tap_bytes"] is None and d["roofline"]["traffic"] is None
    assert 0.2 < c["incloud_fraction"] < 0.4 and c["exact_fp32_cells"] is False          # (the benchmark workload: 0.152)
    assert d["value_as_asked"] > 0 and d["ms_per_step"] > 1.0
