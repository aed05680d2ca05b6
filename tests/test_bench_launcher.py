"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run), on CPU with stand-in ranks:
it starts N processes with the rendezvous variables set, relays rank 0's JSON line alone, and fails -- loudly and with
the failing rank's code -- when any rank fails, terminating the ranks that are still waiting."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_OK = textwrap.dedent('''
    import json, os, sys, time
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    print("noise from rank %d" % r)                       # only rank 0's JSON line may reach the parent's stdout
    if r == 0:
        print(json.dumps({"n_gpus": w, "argv": sys.argv[1:]}))
''')
RANK_FAIL = textwrap.dedent('''
    import os, sys, time
    if os.environ["RANK"] == "1":
        sys.stderr.write("rank 1: no such GPU\\n"); sys.exit(3)
    time.sleep(60)                                        # a rank stuck in the rendezvous: the parent must end it
''')


def run_parent(tmp_path, body, n):
    script = tmp_path / "rank.py"
    script.write_text(body)
    code = ("import sys; sys.path.insert(0, %r); import bench; bench.spawn_ranks(%d, script=%r, argv=['--gpus', '%d', '--x'])"
            % (ROOT, n, str(script), n))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)


def test_parent_relays_rank0_json_only(tmp_path):
    r = run_parent(tmp_path, RANK_OK, 4)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"n_gpus": 4, "argv": ["--gpus", "4", "--x"]}
    assert "noise from rank 2" in r.stderr                # other ranks' stdout goes to stderr


def test_parent_fails_when_a_rank_fails(tmp_path):
    import time
    t0 = time.time()
    r = run_parent(tmp_path, RANK_FAIL, 3)
    assert r.returncode == 3 and r.stdout.strip() == ""
    assert "rank 1 of 3 exited with code 3" in r.stderr
    assert time.time() - t0 < 40                          # the stuck ranks were terminated, not waited for


def test_worker_refuses_a_world_size_mismatch_and_missing_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""
    env.pop("WORLD_SIZE")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "cannot run --gpus 2 here" in r.stderr and r.stdout.strip() == ""     # no GPU in this container
