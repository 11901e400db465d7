"""Pre-flight of the direct-store corner turn (sharded.py: HipFourstep.setup_direct, csrc/fourstep.hip) in a process of its own.

The direct-store form has never run between two PHYSICAL GPUs: its ingredients -- a HIP IPC handle exported on one device and
opened on another, kernels of one GPU storing into the other's memory -- either work or take the process down with a GPU memory
fault, which no error path of the library can catch.  bench.py therefore lets a child of every rank try exactly those ingredients
first; the ranks use the direct-store forms only if every child came back with status 0, and measure with the collective forms
otherwise.  The children find each other through files in a directory the ranks agree on (no process group, no torch).

    python direct_preflight.py <rank> <world> <device> <directory> [kind]     kind: 1 fine-grained, 0 coarse-grained, -1 the library's choice
Exit status: 0 every peer's stores arrived intact; 2 set-up failed (region / export / import: reported, not fatal for the caller);
3 data mismatch; 4 a peer did not show up in time."""
import ctypes
import os
import sys
import time

N = 4096                      # elements every rank stores into every peer's region
WAIT_S = float(os.environ.get("STARKCORE_PREFLIGHT_WAIT_S", "45"))


def _wait_for(paths, what):
    deadline = time.monotonic() + WAIT_S
    while not all(os.path.exists(p) for p in paths):
        if time.monotonic() > deadline:
            sys.stderr.write("direct_preflight: %s did not show up within %.0f s\n" % (what, WAIT_S))
            sys.exit(4)
        time.sleep(0.005)


def _publish(path, data=b"1"):
    with open(path + ".tmp", "wb") as f:
        f.write(data)
    os.replace(path + ".tmp", path)


def pattern(sender, receiver):
    """N canonical residues that name their sender and their receiver"""
    out = bytearray()
    for i in range(N):
        out += ((sender + 1) * 1000003 + (receiver + 1) * 7919 + i).to_bytes(8, "little") + (i * 2654435761 % (1 << 60)).to_bytes(8, "little")
    return bytes(out)


def main():
    rank, world, device, where = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    kind = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    os.environ.setdefault("STARKCORE_NO_TORCH", "1")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import starkcore as sc
    sc.init(device)
    if os.environ.get("STARKCORE_TEST_PREFLIGHT_DIES") == str(rank):      # tests: this rank's child ends the way a GPU memory fault would
        os.abort()
    lib = sc.lib()
    region, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
    if lib.sc_ipc_region_create_ex(16 * N * world, kind, ctypes.byref(region), handle) != 0:
        sys.stderr.write("direct_preflight: rank %d could not create / export its region: %s\n" % (rank, lib.sc_last_error().decode(errors="replace")))
        _publish(os.path.join(where, "handle_%d" % rank), b"")          # (the peers must not wait for it)
        sys.exit(2)
    _publish(os.path.join(where, "handle_%d" % rank), handle.raw)
    _wait_for([os.path.join(where, "handle_%d" % h) for h in range(world)], "a peer's handle")
    peers = {}
    for h in range(world):
        if h == rank:
            continue
        raw = open(os.path.join(where, "handle_%d" % h), "rb").read()
        opened = ctypes.c_void_p()
        if len(raw) != 64 or lib.sc_ipc_region_open(raw, ctypes.byref(opened)) != 0:
            sys.stderr.write("direct_preflight: rank %d could not map rank %d's region: %s\n" % (rank, h, lib.sc_last_error().decode(errors="replace") if len(raw) == 64 else "it has none"))
            _publish(os.path.join(where, "stored_%d" % rank), b"0")
            sys.exit(2)
        peers[h] = opened
    # a KERNEL of this GPU stores into every peer's memory (sc_scale_dev with factor 1: out[i] = in[i] * 1^i), as the column stage does
    one = sc.fe_bytes(1)
    for h, opened in peers.items():
        src = sc.DeviceVector.from_bytes(pattern(rank, h))
        sc._check(lib.sc_scale_dev(src.ptr, ctypes.c_void_p(opened.value + 16 * N * rank), N, one, None))
    own = sc.DeviceVector.from_bytes(pattern(rank, rank))
    sc._check(lib.sc_scale_dev(own.ptr, ctypes.c_void_p(region.value + 16 * N * rank), N, one, None))
    sc.synchronize()
    _publish(os.path.join(where, "stored_%d" % rank))
    _wait_for([os.path.join(where, "stored_%d" % h) for h in range(world)], "a peer's stores")
    if any(open(os.path.join(where, "stored_%d" % h), "rb").read() != b"1" for h in range(world)):
        sys.exit(2)
    mine = sc.DeviceVector.wrap(region.value, N * world, None).to_bytes()
    status = 0
    for h in range(world):
        if mine[16 * N * h:16 * N * (h + 1)] != pattern(h, rank):
            sys.stderr.write("direct_preflight: rank %d does not see what rank %d stored\n" % (rank, h))
            status = 3
    _publish(os.path.join(where, "checked_%d" % rank))
    _wait_for([os.path.join(where, "checked_%d" % h) for h in range(world)], "a peer's check")     # nobody unmaps while a peer still reads
    for opened in peers.values():
        lib.sc_ipc_region_close(opened)
    lib.sc_ipc_region_free(region)
    sys.exit(status)


if __name__ == "__main__":
    main()
