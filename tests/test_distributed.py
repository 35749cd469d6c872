"""The N>1 path on CPU: band partition + framebuffer gather over torch.distributed (gloo, world_size 2 and 3).
The oracle stands in for the renderer: each rank fills its compact band buffer from the oracle's image."""
import os
import socket

import numpy as np
import pytest

import fixtures as fx


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    import ntracer_amd
    from ntracer_amd import distributed as ntd
    import oracle_binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = fx.load("box_n6_1920x1080")
        fmt = ntracer_amd.ImageFormat(w, h, [ntracer_amd.Channel(*c) for c in fx.RGB16], pitch=w * 6 + 8)
        full = ob.OracleScene(6, g["origins"][17], g["axes"][17]).render(w, h, fx.RGB16, pitch=fmt.pitch)
        rows = ntd.owned_rows(h, rank, world)
        compact = torch.from_numpy(np.ascontiguousarray(full[rows]))
        assert compact.numel() == ntd.compact_len(fmt, rank, world)
        img = ntd.gather_framebuffer(compact, fmt, rank, world, dst=0)
        if rank == 0:
            assert np.array_equal(img.numpy(), full)
            open(out_path, "w").write("ok")
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 100), (3, 77), (2, 31)])
def test_band_partition_and_gather(tmp_path, world, h):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(world, _free_port(), 61, h, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_owned_rows_partition_the_image():
    from ntracer_amd import distributed as ntd
    for h in (1, 31, 32, 33, 100, 1080, 4096):
        for world in (1, 2, 3, 4, 8):
            rows = np.concatenate([ntd.owned_rows(h, r, world) for r in range(world)])
            assert sorted(rows.tolist()) == list(range(h))
            sizes = [len(ntd.owned_rows(h, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 32
