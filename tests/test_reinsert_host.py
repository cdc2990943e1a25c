"""The reinsertion passes of the device BVH builder, compiled for the host and run thread by thread
(tools/reinsert_host_check.cpp over tools/hostshim): after every pass the tree is walked -- links consistent, every
node reached once, boxes tight, the marked refit equal to a refit of the whole tree -- and at the end the placement of
triangles and nodes is checked.  No GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("reinsert") / "reinsert_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(ROOT, "tools", "hostshim"), "-I" + os.path.join(ROOT, "pathtracer_amd", "csrc"),
                    os.path.join(ROOT, "tools", "reinsert_host_check.cpp"), "-o", exe], check=True, cwd=ROOT)
    return exe


@pytest.mark.parametrize("n,passes,mod,seed", [(3000, 8, 2, 1), (1500, 8, 1, 3), (8, 4, 2, 2), (9, 4, 1, 1), (2, 2, 1, 1)])
def test_reinsertion_passes_keep_the_tree_valid(checker, n, passes, mod, seed):
    run = subprocess.run([checker, str(n), str(passes), str(mod), str(seed)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout[-2000:] + run.stderr[-2000:]
