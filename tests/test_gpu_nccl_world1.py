"""The RCCL ("nccl") branch of svox_t_amd/parallel.py has to have RUN before the first multi-GPU job (VERDICT r04
"missing" 1b): one rank on cuda:0, every collective forced (tests/nccl_world1_worker.py says what that does and does not
cover), in a child process so that the process group never touches the test runner."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_branch_runs_at_world_size_one(gpu):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_world1_worker.py"), str(port)],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["max_reduce"] == 1.25
    assert r["render_sharded_out_equal"] and r["gather_all_equal"] and r["gather_dst_equal"] and r["cameras_out_equal"]
    assert r["direct_equal"] and r["sparse_equal"]
    # gradients: float atomics reorder sums from run to run -- the tolerance of the repeated-run tests (1e-5 of the largest entry)
    for k in ("render_sharded_grad_maxdiff", "reducer_all_reduce_maxdiff", "reducer_direct_maxdiff", "reducer_touched_maxdiff",
              "cameras_grad_maxdiff"):
        assert r[k] <= 1e-5, (k, r[k])
    assert r["reducer_all_reduce_chunks"] >= 2            # (several row chunks: the chunked form is what ran)
