"""sharded_setup.direct_store_preflight under gloo on a machine WITHOUT a GPU (tests/test_sharded_cpu.py): every rank's child
(stark-anatomy_amd/direct_preflight.py) fails to initialise the library, says so with its exit status, and the ranks agree -- through the
file rendezvous directory rank 0 made and the all-reduce behind it -- that the direct-store forms are not to be tried.  With
PREFLIGHT_FAKE=1 the child is replaced by a stand-in that only walks through the rendezvous (handle / stored / checked files), so that
the agreement on SUCCESS, and on one rank's failure (PREFLIGHT_FAKE_FAILS=<rank>), is covered without a device too."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sharded_setup
    if os.environ.get("PREFLIGHT_FAKE") == "1":
        # the stand-in child: the same files in the same order, no library
        import direct_preflight as real
        stand_in = os.path.join(os.environ["PREFLIGHT_TMP"], "stand_in_%d.py" % rank)
        with open(stand_in, "w") as f:
            f.write("import os, sys\nsys.path.insert(0, %r)\nimport direct_preflight as d\n"
                    "rank, world, where = int(sys.argv[1]), int(sys.argv[2]), sys.argv[4]\n"
                    "if os.environ.get('PREFLIGHT_FAKE_FAILS') == str(rank): os.abort()\n"
                    "for stage in ('handle', 'stored', 'checked'):\n"
                    "    d._publish(os.path.join(where, '%%s_%%d' %% (stage, rank)), b'1' if stage != 'handle' else bytes(64))\n"
                    "    d._wait_for([os.path.join(where, '%%s_%%d' %% (stage, h)) for h in range(world)], stage)\n"
                    "sys.exit(0)\n" % os.path.dirname(real.__file__))
        real_join = os.path.join
        os.path.join = lambda *a: stand_in if a[-1] == "direct_preflight.py" else real_join(*a)      # bench.py builds the child's path with it
    out = sharded_setup.direct_store_preflight(rank, world, torch.device("cpu"), dist, "gloo")
    again = sharded_setup.direct_store_preflight(rank, world, torch.device("cpu"), dist, "gloo")
    assert again is out                                                                    # once per job
    want = os.environ.get("PREFLIGHT_EXPECT", "fail")
    assert out["passed"] is (want == "pass"), out
    if want == "fail" and os.environ.get("PREFLIGHT_FAKE") != "1":
        assert out["this_rank_status"] != 0, out                                           # no GPU here: the child could not initialise the library
    dist.barrier()
    print("ok rank %d: %s" % (rank, out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
