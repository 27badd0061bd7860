"""bench.py --gpus N without a launcher around it: the script starts its own ranks (CPU test of that logic with a
stand-in for the rank program)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def _rank_program(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_ranks_get_their_environment_and_rank0_line_is_relayed(tmp_path, capfd):
    cmd = _rank_program(tmp_path, """
        import json, os, sys
        e = os.environ
        rec = {k: e[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'FASTS_HIP_ID_FILE')}
        open(os.path.join(sys.argv[1], 'env%s.json' % e['RANK']), 'w').write(json.dumps(rec))
        print(json.dumps({"n_gpus": int(e['WORLD_SIZE']), "rank": int(e['RANK'])}))
    """)
    rc = bench.launch_ranks(4, [str(tmp_path)], ngpu=8, command=cmd, timeout=60.)
    assert rc == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"n_gpus": 4, "rank": 0}      # rank 0's line only
    envs = [json.loads((tmp_path / ("env%d.json" % r)).read_text()) for r in range(4)]
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3'] and [e['LOCAL_RANK'] for e in envs] == ['0', '1', '2', '3']
    assert all(e['WORLD_SIZE'] == '4' and e['MASTER_ADDR'] == '127.0.0.1' for e in envs)
    assert len({e['MASTER_PORT'] for e in envs}) == 1 and len({e['FASTS_HIP_ID_FILE'] for e in envs}) == 1
    assert not os.path.exists(envs[0]['FASTS_HIP_ID_FILE'])


def test_a_failing_rank_fails_the_job_and_the_others_are_stopped(tmp_path, capfd):
    cmd = _rank_program(tmp_path, """
        import os, sys, time
        if os.environ['RANK'] == '1':
            sys.exit(3)
        time.sleep(600)
    """)
    rc = bench.launch_ranks(2, [], ngpu=2, command=cmd, timeout=120.)
    assert rc == 3
    assert capfd.readouterr().out == ''


def test_fewer_gpus_than_ranks_is_an_error_not_a_smaller_job(tmp_path, capfd):
    assert bench.launch_ranks(8, [], ngpu=1, command=[sys.executable, '-c', 'print(1)']) != 0
    assert capfd.readouterr().out == ''


def test_command_line_without_launcher_and_without_gpus_exits_nonzero():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'n_gpus' not in r.stdout
