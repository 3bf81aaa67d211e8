"""CPU tests of bench.py's launcher decision (`bench.py --gpus N` without a torchrun environment used to
run ONE GPU silently and print n_gpus: 1) and of its bookkeeping helpers."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_gpus_n_without_torchrun_spawns_n_ranks():
    what, why = bench.launch_plan(8, {}, 8)
    assert what == "spawn" and "8 ranks" in why
    assert bench.launch_plan(2, {}, 8)[0] == "spawn"
    assert bench.launch_plan(1, {}, 8)[0] == "inline"
    assert bench.launch_plan(1, {}, 1)[0] == "inline"


def test_fewer_devices_than_requested_is_an_error():
    what, why = bench.launch_plan(8, {}, 1)
    assert what == "error" and "8" in why and "1 HIP device" in why
    assert bench.launch_plan(0, {}, 8)[0] == "error"


def test_inside_torchrun_the_flag_must_match_the_world():
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "2"}, 8)[0] == "inline"
    assert bench.launch_plan(8, {"WORLD_SIZE": "4"}, 8)[0] == "error"
    assert bench.launch_plan(1, {"WORLD_SIZE": "4"}, 8)[0] == "error"       # (used to pass silently when --gpus was 1)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, 1)[0] == "inline"


def test_the_library_driver_is_one_process():
    assert bench.launch_plan(8, {}, 8, driver="lib")[0] == "inline"
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8, driver="lib")[0] == "error"
    assert bench.launch_plan(8, {}, 4, driver="lib")[0] == "error"


def test_spawn_command_is_a_local_torchrun(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd; seen["env"] = env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_no_device_is_a_loud_exit():
    """No HIP device in this container: bench.py must refuse (the product path has no CPU fallback), not print a line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")


def test_algorithmic_bytes_and_kernel_table():
    b_in, b_out = bench.algorithmic_bytes(200, 500, 8)
    assert b_in + b_out == 992008                                             # DESIGN.md section 2 / SURVEY 8(d)
    tab = bench.kernel_bytes_table(1024, 200, 500, 8)
    assert tab["pass_fused_kernel"] == 1024 * 992008
    for k in ("recursion_pair_kernel", "collapse_miss_kernel", "collapse_wide2_kernel", "meanscan_mfma_kernel"):
        assert k in tab                                                       # the names rocprofv3 prints
    info = bench.host_cpu_info()
    assert info["usable"] >= 1


def test_the_json_line_is_alone_on_stdout_when_a_library_prints_with_c_stdio():
    """RCCL prints a version banner on stdout (C stdio, buffered: it would land BEHIND the JSON line at exit).  While a
    communicator is created bench.py points fd 1 at stderr and flushes C stdio before pointing it back."""
    code = (
        "import sys, ctypes\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "libc = ctypes.CDLL(None)\n"
        "with bench.stdout_to_stderr():\n"
        "    libc.printf(b'RCCL version : banner\\n')\n"
        "print('{\"metric\": 1}', flush=True)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.stdout == '{"metric": 1}\n' and "banner" in r.stderr


def test_secondary_plan_covers_every_config_and_the_scale_run():
    """a default one-GPU run times every other line of the path incl. BASELINE configs[0] and [4]; a default
    N-GPU run (the driver's SCALE launch) times configs[2]'s per-GPU shard and the EM line with its per-iteration all-gather."""
    one = bench.secondary_plan(1, True, False)
    for key in ("b8192", "missing10", "missing10_b8192", "em", "em_missing10", "em_missing10_b8192", "pca", "c4", "c4_em", "c4_missing10",
                "c4_em_missing10", "c1_als", "c5_boot"):
        assert key in one
    eight = bench.secondary_plan(8, True, False)
    assert eight == ["c3", "em", "pass_gather_every_step"]
    cfg, steps, warm, gather = bench.MULTI_SECONDARY["c3"]
    assert cfg["B"] == 8192 and 8 * cfg["B"] == 65536 and gather == "block" and cfg["mode"] == "pass"
    assert bench.MULTI_SECONDARY["pass_gather_every_step"][3] == "step"
    assert bench.secondary_plan(8, False, False) == [] and bench.secondary_plan(1, True, True) == []


def test_gram_flops_are_the_executed_ones():
    """the X'X kernels compute the upper triangle of their tiling; pricing them on 2 T N^2 over-states the
    matrix-pipe fraction by 1.7-1.9x (and could exceed 1)."""
    ex, useful = bench.gram_flops("gram_xx_dma_kernel", 1024, 200, 500)
    assert ex == 2.0 * 500 * 256 * 91 * 1024                     # 13 x 13 tiles of 16 series: 91 upper-triangle tiles
    assert useful == 500.0 * 200 * 201 * 1024
    assert useful < ex < 2.0 * 500 * 200 * 200 * 1024
    exw, _ = bench.gram_flops("gram_xx_wide_kernel", 256, 1000, 2000)
    assert exw == 2.0 * 2000 * 128 * 128 * 36 * 256              # 8 blocks of 128 series: 36 pairs
    assert bench.gram_flops("gram_xx_kernel", 1, 10, 10)[0] == 2000.0


def test_every_default_line_fits_the_hbm_of_one_gpu():
    """the batches a default run keeps resident -- one GPU's secondary lines and the N-GPU plan (BASELINE
    configs[2]: 8192 replicates per GPU) -- with the library's own workspace (dfm_workspace_bytes) stay inside 288 GB per GPU."""
    import pytest
    cfgs = [dict(B=1024, N=200, T=500, r=8, missing=0.0, mode="pass")] + [c for _, c, _, _ in bench.SECONDARY]
    cfgs += [c for c, _, _, _ in bench.MULTI_SECONDARY.values()]
    assert {k for k in bench.secondary_plan(8, True, False)} == set(bench.MULTI_SECONDARY)
    worst = 0
    for cfg in cfgs:
        need = bench.resident_bytes(cfg)
        assert 0 < need < 0.5 * bench.HBM_BYTES_PER_GPU, (cfg, need)           # (half: torch's allocator and RCCL need room too)
        worst = max(worst, need)
    assert worst > 10 * 10**9                                                   # (the 8192-replicate shards are tens of GB: the check is not vacuous)
    big = dict(B=400000, N=200, T=500, r=8, missing=0.0, mode="pass")           # 40 x configs[2]'s shard does NOT fit
    assert bench.resident_bytes(big) > bench.HBM_BYTES_PER_GPU
    with pytest.raises(SystemExit):
        bench.check_distinct_devices([dict(pci_bus_id="0000:05:00.0"), dict(pci_bus_id="0000:05:00.0")], 2)
    assert bench.check_distinct_devices([dict(pci_bus_id="0000:05:00.0"), dict(pci_bus_id="0000:06:00.0")], 2) == 2
    os.environ["DFM_BENCH_ALLOW_SHARED_DEVICE"] = "1"
    try:
        assert bench.check_distinct_devices([dict(pci_bus_id="a"), dict(pci_bus_id="a")], 2) == 1
    finally:
        del os.environ["DFM_BENCH_ALLOW_SHARED_DEVICE"]


def test_em_iteration_bytes():
    """Compulsory bytes of an EM iteration with a k-wide companion state: the panel twice, the parameters in and out."""
    assert bench.em_iteration_bytes(139, 222, 4, 16) == 8 * (2 * 139 * 222 + 2 * (139 * 4 + 139 + 4 * 16 + 16 + 16 + 256) + 1)
    assert bench.em_iteration_bytes(139, 222, 4, 20, 4) - bench.em_iteration_bytes(139, 222, 4, 20) == 8 * 2 * 139 * 4
