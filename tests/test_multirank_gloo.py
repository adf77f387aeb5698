"""The N > 1 path of bench.py on CPU: two ranks under gloo, each owning its own feeds, gathering fixed-size detection
records.  The per-rank detections come from the CPU oracle here (no GPU in this container); on the GPU box the same
pack / gather / unpack code carries the HIP path's outputs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle as O

cfgs = cases.cfgs
feeds = cases.vp.feeds if hasattr(cases.vp, "feeds") else None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _detections(rank, batch, cfg, mm):
    from _vp_pkg import load
    f = load().feeds
    port = O.Port(threads=2)
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    counter = np.zeros((batch, 3), np.int32)
    matches = np.zeros((batch, 22 * mm), np.uint8)
    idx = np.zeros((batch, mm), np.int32)
    for b in range(batch):
        m, i, c = port.frame(cfg.synth(f.feed_seed(rank, b)), cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y,
                             cfg.sampler_mode, cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
        counter[b] = c
        matches[b, : 22 * len(m)] = np.frombuffer(m.tobytes(), np.uint8)
        idx[b, : len(i)] = i.view(np.int32)
    return counter, matches, idx


def _worker(rank, world, port, batch, out_dir):
    import sys
    for p in (cases.ROOT, os.path.join(cases.ROOT, "tests"), os.path.join(cases.ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from _vp_pkg import load
    f = load().feeds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, mm = cfgs.T_RGGB, 64
    counter, matches, idx = _detections(rank, batch, cfg, mm)
    record = torch.zeros((batch, f.record_bytes(mm)), dtype=torch.uint8)
    f.pack_records(torch.from_numpy(counter), torch.from_numpy(matches), torch.from_numpy(idx), record)
    gathered = torch.zeros((world, batch, f.record_bytes(mm)), dtype=torch.uint8)
    f.gather_records(record, gathered, dist)
    gc, gm, gi = f.unpack_records(gathered, mm)
    # every rank sees every rank's detections, bit for bit
    ok = True
    for r in range(world):
        c, m, i = _detections(r, batch, cfg, mm)
        ok &= np.array_equal(gc[r].numpy(), c) and np.array_equal(gm[r].numpy(), m) and np.array_equal(gi[r].numpy(), i)
    # the compact form bench.py gathers (vp_records_compact on the device; here its host restatement): one flat block per rank
    cap = 40
    block = f.pack_compact_host(torch.from_numpy(counter), torch.from_numpy(matches), torch.from_numpy(idx), cap)
    assert block.numel() == f.compact_block_bytes(batch, cap) and block.numel() % 256 == 0
    gblock = torch.zeros((world, block.numel()), dtype=torch.uint8)
    f.gather_records(block, gblock, dist)
    cc, cn, cm, ci = f.unpack_compact(gblock, batch, cap)
    for r in range(world):
        c, m, i = _detections(r, batch, cfg, mm)
        ok &= np.array_equal(cc[r].numpy(), c)
        for b in range(batch):
            n = min(int(c[b, 0]), mm, cap)
            ok &= int(cn[r, b]) == n and cm[r][b].numpy().tobytes() == m[b, :22 * n].tobytes() and np.array_equal(ci[r][b].numpy(), i[b, :n])
    # the overlapped form bench.py uses: double-buffered records, async gathers, wait before a buffer is reused
    recs = [torch.zeros_like(record) for _ in range(2)]
    gath = [torch.zeros_like(gathered) for _ in range(2)]
    pending = [None, None]
    for k in range(5):
        s = k & 1
        if pending[s] is not None:
            pending[s].wait()
            ok &= int(gath[s][:, 0, 0].sum()) == sum(((k - 2) + r) % 251 for r in range(world))
        recs[s].copy_(record)
        recs[s][0, 0] = (k + rank) % 251  # a per-step, per-rank mark
        pending[s] = f.gather_records(recs[s], gath[s], dist, async_op=True)
    for s in range(2):
        pending[s].wait()
    ok &= torch.equal(gath[0][:, 1:], gathered[:, 1:]) and torch.equal(gath[1][:, 1:], gathered[:, 1:])
    # elapsed-time reduction used by bench.py: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t.item()) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"rank{rank}.ok" if ok else f"rank{rank}.bad"), "w").close()


def _packets(rank, batch, cfg, mm, steps, threads):
    """The CPU stage of bench.py --detect for one rank: one detector per feed slot, `steps` consecutive steps (colour model and
    tracker carried along), returns the packets of the last step."""
    from _vp_pkg import load
    vp = load()
    H, f = vp.hostlib, vp.feeds
    dcfg = H.config_defaults()
    dets = []
    for i in range(batch):
        dcfg.cam_id = rank * batch + i
        d = H.Detector(dcfg)
        lines, arcs = cfg.field_markings()
        d.set_geometry_direct(H.nadir_calib(cfg.model()), (cfg.qw, cfg.qh), cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, field=(int(cfg.field_len), int(cfg.field_wid), 300, -1, 10),
                              lines=lines, arcs=arcs)
        dets.append(d)
    counter, matches, _ = _detections(rank, batch, cfg, mm)
    packets = np.zeros((batch, f.PACKET_CAP), np.uint8)
    lengths = np.zeros(batch, np.int32)
    for k in range(1, steps + 1):
        H.detect_batch(dets, matches, counter, mm, k, k / 70.0, k / 70.0, k / 70.0 + 0.002, threads, packets, lengths)
    return dets, packets, lengths


def _worker_packets(rank, world, port, batch, out_dir):
    import sys
    for p in (cases.ROOT, os.path.join(cases.ROOT, "tests"), os.path.join(cases.ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from _vp_pkg import load
    vp = load()
    f = vp.feeds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, mm = cfgs.T_MID, 256
    dets, packets, lengths = _packets(rank, batch, cfg, mm, 2, 2)
    record = f.pack_packets(torch.from_numpy(lengths), torch.from_numpy(packets), torch.zeros((batch, f.packet_record_bytes()), dtype=torch.uint8))
    gathered = torch.zeros((world, batch, f.packet_record_bytes()), dtype=torch.uint8)
    f.gather_records(record, gathered, dist)
    allp = f.unpack_packets(gathered)
    ok = True
    probe = vp.hostlib.Detector()
    for r in range(world):
        _, pk, ln = _packets(r, batch, cfg, mm, 2, 1)  # recomputed single-threaded: thread count must not change a byte
        for i in range(batch):
            ok &= allp[r][i] == pk[i, : ln[i]].tobytes() and ln[i] > 0
            has, src = probe.decode(allp[r][i])
            fr = probe.frame()
            ok &= has and src == 3 and fr["camera_id"] == r * batch + i and fr["frame_number"] == 2
            # the rendered scene of T_MID: robots 0..4 with alternating teams, three balls (one feed also reports a field-line crossing as a ball)
            ok &= sorted(int(b["robot_id"]) for b in fr["robots_yellow"]) == [0, 2, 4] and sorted(int(b["robot_id"]) for b in fr["robots_blue"]) == [1, 3] and len(fr["balls"]) in (3, 4)
    # second step ran on tracked objects: every detector carries tracks of age 2
    for i, d in enumerate(dets):
        tr = d.get_tracked(rank * batch + i)
        ok &= len(tr) in (8, 9) and int(tr["age"].min()) == 2
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"rank{rank}.ok" if ok else f"rank{rank}.bad"), "w").close()


def test_two_ranks_gather_encoded_detection_frames(tmp_path):
    """BASELINE config 4 on CPU: per rank a set of feeds, each with its own detector (main.cpp:319-370 on host threads), the
    encoded SSL_WrapperPackets all-gathered with equal counts; every rank ends up with every feed's packet."""
    world, batch = 2, 2
    mp.spawn(_worker_packets, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["rank0.ok", "rank1.ok"]


def test_two_ranks_gather_detection_records(tmp_path):
    world, batch = 2, 2
    mp.spawn(_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["rank0.ok", "rank1.ok"]


def test_record_layout_roundtrip():
    from _vp_pkg import load
    f = load().feeds
    mm, B = 2000, 3
    assert f.record_bytes(mm) % 16 == 0 and f.record_bytes(mm) >= 16 + 26 * mm
    rng = np.random.default_rng(0)
    counter = torch.from_numpy(rng.integers(0, 5000, (B, 3), dtype=np.int32))
    matches = torch.from_numpy(rng.integers(0, 256, (B, 22 * mm), dtype=np.uint8))
    idx = torch.from_numpy(rng.integers(0, 1 << 21, (B, mm), dtype=np.int32))
    rec = f.pack_records(counter, matches, idx, torch.zeros((B, f.record_bytes(mm)), dtype=torch.uint8))
    c, m, i = f.unpack_records(rec[None], mm)
    assert torch.equal(c[0], counter) and torch.equal(m[0], matches) and torch.equal(i[0], idx)
    assert len({f.feed_seed(r, b) for r in range(8) for b in range(64)}) == 8 * 64
