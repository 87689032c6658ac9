"""run a chain test of tests/test_iteration_gpu.py against a recorded device run (tests/_replay.py), without a GPU:

    THX_CHAIN_DUMP=gpurun_out/chain python -m pytest tests/test_iteration_gpu.py -m gpu -k point_group     # on the GPU box
    python tools/probes/replay_chain.py gpurun_out/chain test_iteration_matches_oracle_chain_with_point_group C4 160   # anywhere

arguments after the test's name are its parameters (int / float / None where they parse as such)."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests")]
os.environ["THX_CHAIN_REPLAY"] = os.path.abspath(sys.argv[1])


def parse(a):
    if a == "None":
        return None
    for f in (int, float):
        try:
            return f(a)
        except ValueError:
            pass
    return a


import test_iteration_gpu as Tm   # noqa: E402
from oracle import oracle as O    # noqa: E402

O.lib()
getattr(Tm, sys.argv[2])(O, None, *[parse(a) for a in sys.argv[3:]])
print("replay: passed")
