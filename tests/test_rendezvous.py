"""fleet_bench's ranks find rank 0's RCCL id through a file named after the job (host/wg_rendezvous.hpp).  CPU test of that
protocol alone: a well-formed file a previous job left under this job's name is NOT consumed (its writer is no longer alive),
the blob rank 0 publishes afterwards is -- by ranks started through per-rank wrappers that do not exec (every rank a different
parent) and were already polling; a file of another world size, a finished job's file and a torn file are not taken; the name
comes from the launcher's environment (port + run id / PMIx namespace / SLURM job), the parent's pid only when there is none."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_stale_id_file_of_a_previous_job_is_not_consumed(tmp_path):
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_rendezvous")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "jrl-walkgen_amd"), "bin/test_rendezvous"])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rendezvous ok" in r.stdout
    assert "written by a process that is gone" in r.stderr       # the stale file was seen and passed over
