"""Multi-GPU sharding of the hot path: independent camera feeds, one (set of) feed(s) per rank, no data-path collective.

The only exchange the path has is at the end of a step: every rank contributes one fixed-size detection record per frame
(counter[3], the 22-byte match records, their pixel indices) and all ranks gather them.  The reference has
no counterpart (one process per camera exchanging protobufs over UDP multicast, udpsocket.cpp:94-100); on one node the
gather is a single RCCL all-gather of ~52 KB per frame, latency-bound.

Device-agnostic torch code so the same functions run under `gloo` on CPU tensors in the tests."""
from __future__ import annotations

import torch

# Frames per step and GPU of the benchmark launch (bench.py --batch default; tests/test_gpu_round3.py holds this very launch to the oracle).  One launch
# group of 256 cfg2 feeds: 0.8 GB of sensor frames, 3.2 GB of flat, 3.2 GB of table -- sized for 288 GB of HBM, not for a camera count; measured
# (profiles/r05_batch_sweep.json): 128 -> 72.7 k frames/s, 256 -> 75.8 k, 512 -> 76.1 k (the SAT pass: 0.665 / 0.696 / 0.672 of 8 TB/s).
BENCH_FRAMES_PER_STEP = 256

MATCH_BYTES = 22
HEADER_BYTES = 16  # counter[3] as int32 + 4 bytes padding


def record_bytes(max_matches: int) -> int:
    n = HEADER_BYTES + MATCH_BYTES * max_matches + 4 * max_matches
    return n + (-n) % 16


def feed_seed(rank: int, frame: int) -> int:
    """Synthetic feed of rank `rank`, frame `frame`: distinct scenes for every (rank, frame)."""
    return rank * 100000 + frame


def pack_records(counter: torch.Tensor, matches: torch.Tensor, idx: torch.Tensor, record: torch.Tensor) -> torch.Tensor:
    """counter (B,3) int32, matches (B, 22*max) uint8, idx (B, max) int32 -> record (B, record_bytes) uint8, in place."""
    b = counter.shape[0]
    mm = idx.shape[1]
    record[:, 0:12] = counter.contiguous().view(torch.uint8).view(b, 12)
    record[:, HEADER_BYTES:HEADER_BYTES + MATCH_BYTES * mm] = matches
    record[:, HEADER_BYTES + MATCH_BYTES * mm:HEADER_BYTES + (MATCH_BYTES + 4) * mm] = idx.contiguous().view(torch.uint8).view(b, 4 * mm)
    return record


def unpack_records(gathered: torch.Tensor, max_matches: int):
    """gathered (world, B, record_bytes) uint8 -> (counter (world,B,3) int32, matches (world,B,22*max) uint8, idx (world,B,max) int32)."""
    w, b, _ = gathered.shape
    mm = max_matches
    counter = gathered[:, :, 0:12].contiguous().view(torch.int32).view(w, b, 3)
    matches = gathered[:, :, HEADER_BYTES:HEADER_BYTES + MATCH_BYTES * mm].contiguous()
    idx = gathered[:, :, HEADER_BYTES + MATCH_BYTES * mm:HEADER_BYTES + (MATCH_BYTES + 4) * mm].contiguous().view(torch.int32).view(w, b, mm)
    return counter, matches, idx


# ---- compact records: what vp_records_compact (include/vp_hip.h) leaves per feed -- counter[3], n and the n records the host loop reads -------------
COMPACT_CAP = 256  # records per slot in bench.py (the synthetic scenes give ~130; the list capacity max_matches stays 2000)


def compact_slot_bytes(cap: int = COMPACT_CAP) -> int:
    """vp_record_layout_compact: [counter 3 x i32 | n i32 | cap x 22 B | pad to 4 | cap x u32], padded to 16 bytes."""
    m = (16 + MATCH_BYTES * cap + 3) & ~3
    return (m + 4 * cap + 15) & ~15


def compact_block_bytes(batch: int, cap: int = COMPACT_CAP) -> int:
    return (compact_slot_bytes(cap) * batch + 255) & ~255


def pack_compact_host(counter: torch.Tensor, matches: torch.Tensor, idx: torch.Tensor, cap: int = COMPACT_CAP) -> torch.Tensor:
    """Host restatement of vp_records_compact (CPU tensors; tests and the unpacker's inverse): counter (B,3) int32, matches (B, 22*max) uint8,
    idx (B, max) int32 -> one block (compact_block_bytes,) uint8.  Bytes of a slot beyond its n records stay zero here (the device leaves them as they were)."""
    b, mm = counter.shape[0], idx.shape[1]
    slot = compact_slot_bytes(cap)
    off_idx = (16 + MATCH_BYTES * cap + 3) & ~3
    out = torch.zeros(compact_block_bytes(b, cap), dtype=torch.uint8)
    for f in range(b):
        n = max(0, min(int(counter[f, 0]), mm, cap))
        base = f * slot
        head = torch.tensor([int(counter[f, 0]), int(counter[f, 1]), int(counter[f, 2]), n], dtype=torch.int32)
        out[base:base + 16] = head.view(torch.uint8)
        out[base + 16:base + 16 + MATCH_BYTES * n] = matches[f, :MATCH_BYTES * n]
        out[base + off_idx:base + off_idx + 4 * n] = idx[f, :n].contiguous().view(torch.uint8)
    return out


def unpack_compact(gathered: torch.Tensor, batch: int, cap: int = COMPACT_CAP):
    """gathered (world, compact_block_bytes) uint8 -> (counter (world,B,3) int32, n (world,B) int32, matches: list[world][B] of (n, 22) uint8,
    idx: list[world][B] of (n,) int32)."""
    g = gathered.cpu()
    w = g.shape[0]
    slot = compact_slot_bytes(cap)
    off_idx = (16 + MATCH_BYTES * cap + 3) & ~3
    slots = g[:, :slot * batch].reshape(w, batch, slot)
    head = slots[:, :, :16].contiguous().view(torch.int32).view(w, batch, 4)
    counter, n = head[:, :, :3].contiguous(), head[:, :, 3].contiguous()
    matches = [[slots[r, f, 16:16 + MATCH_BYTES * int(n[r, f])].reshape(-1, MATCH_BYTES) for f in range(batch)] for r in range(w)]
    idx = [[slots[r, f, off_idx:off_idx + 4 * int(n[r, f])].contiguous().view(torch.int32) for f in range(batch)] for r in range(w)]
    return counter, n, matches, idx


def gather_records(record: torch.Tensor, gathered: torch.Tensor, dist, async_op: bool = False):
    """All-gather the per-rank record block: one collective per step, equal counts on every rank.
    async_op=True returns the work handle: the collective then runs beside the next step's kernels (RCCL: its own stream);
    the caller waits on the handle before it reuses `record` / reads `gathered`."""
    # flattened views: (world * B, bytes) <- (B, bytes); the gloo backend insists on the concatenated shape, RCCL accepts it too
    if record.dim() == 1:  # one flat block per rank (compact records)
        work = dist.all_gather_into_tensor(gathered.view(-1), record, async_op=async_op)
        return work if async_op else gathered
    work = dist.all_gather_into_tensor(gathered.view(-1, gathered.shape[-1]), record, async_op=async_op)
    return work if async_op else gathered


# ---- encoded detection frames (SSL_WrapperPacket bytes per feed, BASELINE config 4) -------------------------------------------
PACKET_CAP = 4092  # bytes per feed and step: 16 robots x 45 B + 8 balls x 27 B + header is < 1 KB; a UDP datagram of the reference is < 64 KB


def packet_record_bytes(cap: int = PACKET_CAP) -> int:
    return 4 + cap  # int32 length + payload


def pack_packets(lengths: torch.Tensor, packets: torch.Tensor, record: torch.Tensor) -> torch.Tensor:
    """lengths (B,) int32, packets (B, cap) uint8 -> record (B, 4 + cap) uint8, in place."""
    b = lengths.shape[0]
    record[:, 0:4] = lengths.contiguous().view(torch.uint8).view(b, 4)
    record[:, 4:] = packets
    return record


def unpack_packets(gathered: torch.Tensor):
    """gathered (world, B, 4 + cap) uint8 -> list (per rank) of lists (per feed) of bytes: the SSL_WrapperPacket each feed emitted."""
    g = gathered.cpu()
    w, b, _ = g.shape
    lengths = g[:, :, 0:4].contiguous().view(torch.int32).view(w, b)
    out = []
    for r in range(w):
        row = []
        for i in range(b):
            n = int(lengths[r, i])
            row.append(bytes(g[r, i, 4:4 + n].numpy().tobytes()) if n > 0 else b"")
        out.append(row)
    return out
