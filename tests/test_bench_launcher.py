"""bench.py --gpus N without an external launcher: the set of child-process environments it builds (CPU only: nothing here
starts a rank or touches a GPU)."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(HERE), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_environments_describe_one_process_per_gpu():
    bench = load_bench()
    for n in (2, 4, 8):
        envs = bench.rank_environments(n, base_env={"PATH": "/usr/bin", "VR_EXP_FLAVOUR": "6"}, port=23456)
        assert len(envs) == n
        assert [e["RANK"] for e in envs] == [str(r) for r in range(n)]
        assert [e["LOCAL_RANK"] for e in envs] == [str(r) for r in range(n)]      # rank r renders on device r
        assert all(e["WORLD_SIZE"] == str(n) and e["LOCAL_WORLD_SIZE"] == str(n) for e in envs)
        assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "23456" for e in envs)
        assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)          # RCCL across processes needs dmabuf IPC
        assert all(e["VR_BENCH_SPAWNED"] == "1" and e["PATH"] == "/usr/bin" and e["VR_EXP_FLAVOUR"] == "6" for e in envs)
    # an environment that already says otherwise is kept (the pool exports HSA_ENABLE_IPC_MODE_LEGACY itself)
    envs = bench.rank_environments(2, base_env={"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, port=1)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" for e in envs)


def test_rank_environments_pick_a_free_port():
    bench = load_bench()
    envs = bench.rank_environments(3, base_env={})
    ports = {e["MASTER_PORT"] for e in envs}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_launcher_decision_needs_no_gpu_call():
    """`--gpus N` with no WORLD_SIZE must branch to launch_ranks before torch is imported: the parent process never initialises a
    GPU (the ranks are started as fresh child processes).  Checked on the source: the branch sits above `import torch`."""
    src = open(os.path.join(os.path.dirname(HERE), "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args.gpus") < main.index("import torch")
