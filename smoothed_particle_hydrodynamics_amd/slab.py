"""1-D slab decomposition of the FULL-mode step across GPUs.

The reference is single-process; this is the multi-GPU counterpart of its
`SPH::run()` loop.  One slab per GPU owns a range of global z-planes of the FULL cell grid
(include/sph_hip.h, "multi-GPU" section).  Per step each slab packs one message per neighbour
on the device, the transport moves the two messages (RCCL send/recv over xGMI through
`torch.distributed`, backend "nccl"; or plain pointer hand-over between slabs that live in one
process), the slab unpacks what it received and steps.  There is no collective in the data
path: each slab talks to its two neighbours only.

The orchestration here is backend-agnostic: `HipSlab` drives the HIP library; the tests plug a
CPU stand-in with the same five methods to exercise planning, message flow and the
torch.distributed transport under gloo.
"""
import ctypes as C

import numpy as np

from .lib import SphHipError, load_library

HALO = 2
HEADER_BYTES = 32
RECORD_BYTES = 32


# ---- planning ---------------------------------------------------------------------------------

def plane_of(params, z):
    """Global FULL-grid z-plane of each z coordinate — same arithmetic as the device
    (fp32 multiply, floor, clamp)."""
    z = np.asarray(z, np.float32)
    c = np.floor(z * np.float32(params.full_cell_inv))
    c = np.where(np.isfinite(c), c, -1.0)
    return np.clip(c, 0, params.full_cells_z - 1).astype(np.int64)


def cuts_from_histogram(hist, world, min_planes=2 * HALO):
    """Cut planes [c0=0, c1, ..., c_world=nz] that balance the particle count per slab, from the
    number of particles per z-plane.  Every slab gets at least `min_planes` planes (a slab must
    be at least as thick as the two halos it feeds).  Deterministic."""
    hist = np.asarray(hist, np.float64)
    nz = hist.size
    if world * min_planes > nz:
        raise ValueError("grid has %d planes: too few for %d slabs of >= %d planes" %
                         (nz, world, min_planes))
    cum = np.concatenate([[0.0], np.cumsum(hist)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(cum, target, side="left"))
        c = max(c, cuts[-1] + min_planes)
        c = min(c, nz - (world - r) * min_planes)
        cuts.append(c)
    cuts.append(nz)
    return cuts


def plan_cuts(params, z, world, min_planes=2 * HALO):
    """cuts_from_histogram of the z coordinates `z`: every rank computes the same cuts from the
    same z."""
    hist = np.bincount(plane_of(params, z), minlength=params.full_cells_z)
    return cuts_from_histogram(hist, world, min_planes)


def message_bytes(capacity_records):
    return HEADER_BYTES + capacity_records * RECORD_BYTES


# ---- one slab on one GPU ------------------------------------------------------------------------

def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipSlab:
    """A slab context of libsph_hip.so (sph_hip_create_slab) plus its message buffers."""

    def __init__(self, params, plane_lo, plane_hi, capacity, msg_capacity, device=0,
                 has_left=True, has_right=True, stream=None):
        import torch
        self._torch = torch
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self.params = params.copy()
        self.plane_lo, self.plane_hi = int(plane_lo), int(plane_hi)
        self.capacity, self.msg_capacity = int(capacity), int(msg_capacity)
        # records a message may hold right now (<= msg_capacity, what the buffers can take): what
        # the pack kernels enforce and what the transport moves - see trim_messages()
        self.msg_active = self.msg_capacity
        self._settings = {}
        self._counts_host = None
        self.device = torch.device("cuda", device)
        rc = self._lib.sph_hip_create_slab(C.byref(self._ctx), C.byref(self.params), self.capacity,
                                           int(device), self.plane_lo, self.plane_hi)
        if rc != 0:
            msg = self._lib.sph_hip_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise SphHipError("sph_hip_create_slab failed (%d): %s" % (rc, msg))
        # All launches of this slab go to ONE torch stream (a real stream object: torch's default
        # stream has the NULL handle, which the C ABI reads as "use the context's own stream").
        # torch.distributed orders its RCCL calls against the stream that is current when they
        # are issued, so the transport runs under `with torch.cuda.stream(self.stream)`; slabs
        # sharing a process share the stream, which orders pack(r) before unpack(r+1).
        with torch.cuda.device(self.device):
            self.stream = stream if stream is not None else torch.cuda.Stream()
        self._check(self._lib.sph_hip_set_stream(self._ctx, C.c_void_p(self.stream.cuda_stream)),
                    "set_stream")
        nbytes = message_bytes(self.msg_capacity)
        with torch.cuda.stream(self.stream):
            mk = lambda on: (torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
                             if on else None)
            self.send_left, self.send_right = mk(has_left), mk(has_right)
            self.recv_left, self.recv_right = mk(has_left), mk(has_right)
        self.stream.synchronize()

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.sph_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.sph_hip_last_error(self._ctx).decode()
            raise SphHipError("%s failed (%d): %s" % (what, rc, msg))

    def upload(self, ids, pos, vel, mass, all_masses_equal):
        ids = np.ascontiguousarray(ids, np.uint32)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1)
        mass = np.ascontiguousarray(mass, np.float32)
        self.all_masses_equal = bool(all_masses_equal)
        self._check(self._lib.sph_hip_slab_upload(self._ctx, ids.size, _ptr(pos), _ptr(vel),
                                                  _ptr(mass), _ptr(ids), int(all_masses_equal)),
                    "sph_hip_slab_upload")

    @staticmethod
    def _dp(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def pack(self):
        self._check(self._lib.sph_hip_slab_pack(self._ctx, self._dp(self.send_left),
                                                self._dp(self.send_right), self.msg_active),
                    "sph_hip_slab_pack")

    def unpack(self, recv_left=None, recv_right=None):
        left = recv_left if recv_left is not None else self.recv_left
        right = recv_right if recv_right is not None else self.recv_right
        self._check(self._lib.sph_hip_slab_unpack(self._ctx, self._dp(left), self._dp(right),
                                                  self.msg_active), "sph_hip_slab_unpack")

    def step(self):
        self._check(self._lib.sph_hip_step(self._ctx), "sph_hip_step")

    def step_begin(self, exchange_stream=None):
        """Cell build, density, acceleration of the planes next to a neighbour, messages packed
        into send_left / send_right (sph_hip_slab_step_begin).  exchange_stream: torch stream the
        transport will run on; the border work is enqueued there (None: the slab's stream)."""
        ptr = C.c_void_p(exchange_stream.cuda_stream) if exchange_stream is not None else None
        self._check(self._lib.sph_hip_slab_step_begin(self._ctx, self._dp(self.send_left),
                                                      self._dp(self.send_right), self.msg_active,
                                                      ptr),
                    "sph_hip_slab_step_begin")

    def step_end(self):
        """Acceleration of the interior, integrate (sph_hip_slab_step_end)."""
        self._check(self._lib.sph_hip_slab_step_end(self._ctx), "sph_hip_slab_step_end")

    # ---- native RCCL exchange (the library issues ncclSend/ncclRecv itself) ----------------------
    def comm_init(self, unique_id, rank, world):
        """Create this slab's RCCL communicator from the 128-byte id every rank shares
        (rccl_unique_id() on one rank, then any broadcast)."""
        buf = (C.c_char * len(unique_id)).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.sph_hip_slab_comm_init(self._ctx, buf, len(unique_id), int(rank),
                                                     int(world), self.msg_capacity),
                    "sph_hip_slab_comm_init")

    def comm_run(self, steps):
        """`steps` steps with the overlapped neighbour exchange, all enqueued by the library."""
        self._check(self._lib.sph_hip_slab_comm_run(self._ctx, int(steps)), "sph_hip_slab_comm_run")

    def comm_trim(self, slack=1.25, extra=1024):
        """Native exchange: agree on the message size from what was packed last
        (sph_hip_slab_comm_trim; collective, synchronises)."""
        n = C.c_int32()
        self._check(self._lib.sph_hip_slab_comm_trim(self._ctx, float(slack), int(extra), C.byref(n)),
                    "sph_hip_slab_comm_trim")
        self.msg_active = n.value
        return n.value

    def comm_selftest(self):
        self._check(self._lib.sph_hip_slab_comm_selftest(self._ctx), "sph_hip_slab_comm_selftest")

    def comm_exchange_check(self):
        """One checked message to and from each neighbour through the native calls
        (sph_hip_slab_comm_exchange_check; after comm_init, before the first step)."""
        self._check(self._lib.sph_hip_slab_comm_exchange_check(self._ctx), "sph_hip_slab_comm_exchange_check")

    def comm_stats(self):
        """dict: records the messages are transferred with / allocated for, growths, steps run
        (sph_hip_slab_comm_stats)."""
        out = (C.c_int32 * 4)()
        self._check(self._lib.sph_hip_slab_comm_stats(self._ctx, out), "sph_hip_slab_comm_stats")
        return {"active_records": out[0], "capacity_records": out[1], "growths": out[2], "steps": out[3]}

    def synchronize(self):
        self._check(self._lib.sph_hip_synchronize(self._ctx), "sph_hip_synchronize")

    def send_counts(self):
        """Records in the two messages packed last (synchronises the slab's stream)."""
        self.stream.synchronize()
        out = []
        for m in (self.send_left, self.send_right):
            out.append(int(m[:4].view(self._torch.int32)[0].item()) if m is not None else 0)
        return tuple(out)

    def export_records(self):
        """The owned particles as message records, float32 [n, 8] = {x,y,z,m,vx,vy,vz,id bits}, in a
        tensor on the slab's device (sph_hip_slab_export_records): the state never visits the host."""
        torch = self._torch
        with torch.cuda.device(self.device):
            buf = torch.empty((self.capacity, 8), dtype=torch.float32, device=self.device)
        rows = C.c_int32()
        self._check(self._lib.sph_hip_slab_export_records(self._ctx, C.c_void_p(buf.data_ptr()),
                                                          self.capacity, C.byref(rows)),
                    "sph_hip_slab_export_records")
        return buf[:rows.value]

    def upload_records(self, records, all_masses_equal):
        """This slab's owned particles from float32 [n, 8] message records on its device
        (sph_hip_slab_upload_records); the caller has ordered them by persistent id."""
        torch = self._torch
        rec = records.contiguous()
        assert rec.dtype == torch.float32 and rec.device == self.device and rec.shape[-1] == 8
        self.all_masses_equal = bool(all_masses_equal)
        self._torch.cuda.current_stream(self.device).synchronize()   # the records were made on torch's stream
        self._check(self._lib.sph_hip_slab_upload_records(self._ctx, C.c_void_p(rec.data_ptr()),
                                                          int(rec.shape[0]), int(all_masses_equal)),
                    "sph_hip_slab_upload_records")

    def download_mass(self):
        """Masses of the owned particles, in the row order of download()."""
        cap = self.capacity
        mass = np.zeros(cap, np.float32)
        rows = C.c_int32()
        self._check(self._lib.sph_hip_slab_download_mass(self._ctx, cap, C.byref(rows), _ptr(mass)),
                    "sph_hip_slab_download_mass")
        return mass[:rows.value]

    def poll_errors(self):
        """Non-blocking look at the exchange's error bits (sph_hip_slab_poll_errors): raises
        SphHipError once a copy of a non-zero error word has reached the host; otherwise asks for
        the next copy and returns the bits seen so far (0)."""
        e = C.c_int32()
        self._check(self._lib.sph_hip_slab_poll_errors(self._ctx, C.byref(e)),
                    "sph_hip_slab_poll_errors")
        return e.value

    def status(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.sph_hip_slab_status(self._ctx, C.byref(a), C.byref(b), C.byref(c)),
                    "sph_hip_slab_status")
        return dict(live=a.value, owned=b.value, errors=c.value)

    def download(self):
        rows = C.c_int32()
        cap = self.capacity
        out = dict(ids=np.zeros(cap, np.uint32), pos=np.zeros(3 * cap, np.float32),
                   vel=np.zeros(3 * cap, np.float32), rho=np.zeros(cap, np.float32),
                   acc=np.zeros(3 * cap, np.float32), ncount=np.zeros(cap, np.int32))
        self._check(self._lib.sph_hip_slab_download(self._ctx, cap, C.byref(rows), _ptr(out["ids"]),
                                                    _ptr(out["pos"]), _ptr(out["vel"]),
                                                    _ptr(out["rho"]), _ptr(out["acc"]),
                                                    _ptr(out["ncount"])), "sph_hip_slab_download")
        n = rows.value
        return dict(ids=out["ids"][:n], pos=out["pos"][:3 * n], vel=out["vel"][:3 * n],
                    rho=out["rho"][:n], acc=out["acc"][:3 * n], ncount=out["ncount"][:n])

    def tile_stats(self):
        """dict of the last step's LDS-tile statistics (sph_hip_get_tile_stats), as SPH.tileStats()."""
        out = (C.c_int32 * 20)()
        self._check(self._lib.sph_hip_get_tile_stats(self._ctx, C.byref(out)), "sph_hip_get_tile_stats")
        v = list(out)
        return {"over_level": v[0:12], "workgroups": v[12], "largest_tile": v[13],
                "untiled_density": v[14], "untiled_acceleration": v[15],
                "capacity_density": v[16], "capacity_acceleration": v[17], "wide_entries": v[18],
                "list_capacity": v[19]}

    def phase_totals(self):
        ms = (C.c_double * 6)()
        k = C.c_int32()
        self._check(self._lib.sph_hip_get_phase_totals(self._ctx, C.byref(ms), C.byref(k)),
                    "sph_hip_get_phase_totals")
        return list(ms), k.value

    def set_timing(self, level):
        self._check(self._lib.sph_hip_set_timing(self._ctx, int(level)), "sph_hip_set_timing")
        self._settings["timing"] = int(level)

    def set_timing_stride(self, every):
        self._check(self._lib.sph_hip_set_timing_stride(self._ctx, int(every)),
                    "sph_hip_set_timing_stride")
        self._settings["timing_stride"] = int(every)

    def set_arithmetic(self, arithmetic):
        """ARITH_EXACT / ARITH_FAST pair arithmetic (sph_hip_set_arithmetic); slabs start exact."""
        self._check(self._lib.sph_hip_set_arithmetic(self._ctx, int(arithmetic)), "sph_hip_set_arithmetic")
        self._settings["arithmetic"] = int(arithmetic)

    def settings(self):
        """What set_timing / set_timing_stride / set_arithmetic were last given: a slab that
        replaces this one (DistSlabStepper.rebalance) is set up alike."""
        return dict(self._settings)

    def apply_settings(self, settings):
        if "arithmetic" in settings:
            self.set_arithmetic(settings["arithmetic"])
        if "timing" in settings:
            self.set_timing(settings["timing"])
        if "timing_stride" in settings:
            self.set_timing_stride(settings["timing_stride"])

    def poll_send_counts(self):
        """Records in the two messages, without draining the stream: returns what the copy
        requested by the PREVIOUS call brought ((left, right), or None the first time) and
        requests the next asynchronous copy of the two header words into pinned host memory."""
        torch = self._torch
        if self._counts_host is None:
            self._counts_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._counts_event = torch.cuda.Event()
            self._counts_pending = False
        seen = None
        if self._counts_pending:
            self._counts_event.synchronize()          # requested one polling interval ago
            seen = (int(self._counts_host[0]), int(self._counts_host[1]))
        with torch.cuda.stream(self.stream):
            for k, m in enumerate((self.send_left, self.send_right)):
                if m is not None:
                    self._counts_host[k:k + 1].copy_(m[:4].view(torch.int32), non_blocking=True)
            self._counts_event.record(self.stream)
        self._counts_pending = True
        return seen

    def reset_timings(self):
        self._check(self._lib.sph_hip_reset_timings(self._ctx), "sph_hip_reset_timings")

    def energy(self):
        ke, pe = C.c_float(), C.c_float()
        self._check(self._lib.sph_hip_get_energy(self._ctx, C.byref(ke), C.byref(pe)),
                    "sph_hip_get_energy")
        return ke.value, pe.value


# ---- orchestration ------------------------------------------------------------------------------

def split_scene(params, cuts, rank, pos, vel, mass):
    """The particles (with their global ids) that slab `rank` owns initially."""
    pl = plane_of(params, np.asarray(pos, np.float32).reshape(-1, 3)[:, 2])
    mine = np.nonzero((pl >= cuts[rank]) & (pl < cuts[rank + 1]))[0]
    pos3 = np.asarray(pos, np.float32).reshape(-1, 3)
    vel3 = np.asarray(vel, np.float32).reshape(-1, 3)
    return (mine.astype(np.uint32), np.ascontiguousarray(pos3[mine]).reshape(-1),
            np.ascontiguousarray(vel3[mine]).reshape(-1), np.ascontiguousarray(mass[mine]))


def slab_capacities(counts_per_plane, cuts, rank, slack=1.5):
    """(entry capacity of slab `rank`, message capacity) from the initial plane histogram.

    The message capacity is the same on every rank (a sender's and its receiver's buffers must
    have one size): it covers the fullest halo strip next to any cut."""
    nz = counts_per_plane.size
    lo, hi = cuts[rank], cuts[rank + 1]
    held = counts_per_plane[max(lo - HALO, 0):min(hi + HALO, nz)].sum()
    cap = int(held * slack) + 4096
    strip = 0
    for c in cuts[1:-1]:
        strip = max(strip, counts_per_plane[max(c - HALO, 0):c].sum(),
                    counts_per_plane[c:min(c + HALO, nz)].sum())
    msg = int(strip * slack * 1.5) + 4096
    return cap, msg


def rccl_unique_id():
    """128 bytes naming a new RCCL communicator (call on ONE rank, broadcast to the others)."""
    lib = load_library()
    buf = (C.c_char * 128)()
    rc = lib.sph_hip_rccl_unique_id(buf, 128)
    if rc != 0:
        raise SphHipError("sph_hip_rccl_unique_id failed (%d): %s" %
                          (rc, lib.sph_hip_last_error(None).decode()))
    return bytes(buf)


class NativeSlabStepper:
    """The per-rank loop with the exchange issued by libsph_hip.so itself (RCCL through dlopen):
    torch.distributed is used once, to hand rank 0's communicator id to everybody."""

    def __init__(self, slab, rank, world, group=None):
        import torch.distributed as dist
        ident = [rccl_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        slab.comm_init(ident[0], rank, world)
        self.slab = slab
        self.rebalances = 0          # (the native loop keeps its cuts: rebalancing is DistSlabStepper's)

    def step(self):
        self.slab.comm_run(1)

    def run(self, steps):
        self.slab.comm_run(steps)

    def trim_messages(self, slack=1.25, extra=1024):
        return self.slab.comm_trim(slack, extra)

    @property
    def message_growths(self):
        """times the library put trimmed messages back to their allocated size (it does so before
        they overflow, by itself: sph_hip_slab_comm_stats)"""
        return self.slab.comm_stats()["growths"]


def neighbour_exchange_works(rank, world, device, group=None, nbytes=4096, timeout_s=60.0):
    """Pre-flight of the slab exchange: one small batch_isend_irecv with each neighbouring rank on
    `device` ("cuda" for RCCL, "cpu" for gloo), checked for content.  Returns this rank's verdict:
    an exception counts as "no", and so does an exchange that has not completed after `timeout_s`
    (waited for by a helper thread: a transfer that hangs must not hang the caller, who then agrees on
    a fallback over a group that does not need the path under test - bench.py runs this in a
    helper process, so that a hung RCCL call can be left behind).  Every rank must call it."""
    import torch
    import torch.distributed as dist
    try:
        mine = torch.full((nbytes,), (rank + 1) & 0xff, dtype=torch.uint8, device=device)
        from_left, from_right = torch.zeros_like(mine), torch.zeros_like(mine)
        ops = []
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, mine, rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, from_left, rank - 1, group))
        if rank + 1 < world:
            ops.append(dist.P2POp(dist.isend, mine, rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, from_right, rank + 1, group))
        # the waiting is done by a helper thread: a transfer that never completes leaves that
        # thread behind (daemon), not the caller
        import threading
        finished, failure = threading.Event(), []
        on_device = str(device).startswith("cuda")

        def wait_for_it():
            try:
                if on_device:
                    torch.cuda.set_device(mine.device)
                for req in (dist.batch_isend_irecv(ops) if ops else []):
                    req.wait()
                if on_device:
                    torch.cuda.synchronize()
            except Exception as exc:      # noqa: BLE001
                failure.append(exc)
            finished.set()

        threading.Thread(target=wait_for_it, daemon=True).start()
        if not finished.wait(timeout_s):
            raise TimeoutError("no completion after %.0f s" % timeout_s)
        if failure:
            raise failure[0]
        ok = True
        if rank > 0:
            ok = ok and bool((from_left == (rank & 0xff)).all().item())
        if rank + 1 < world:
            ok = ok and bool((from_right == ((rank + 2) & 0xff)).all().item())
        return ok
    except Exception as exc:      # noqa: BLE001 - whatever the backend raises means "does not work"
        import sys
        print("slab exchange pre-flight failed on rank %d: %r" % (rank, exc), file=sys.stderr, flush=True)
        return False


class DistTransport:
    """Neighbour exchange over torch.distributed point-to-point ops (RCCL when the backend is
    "nccl": each slab pair has its own xGMI link).  One batch per step: send left/right,
    receive left/right.

    exchange() orders the batch on the slab's own stream.  begin()/finish() put it on the
    communication stream (comm_stream(): the stream handed to HipSlab.step_begin, which enqueues
    the packing there), so that whatever the slab enqueues on its own stream between the two calls
    runs concurrently with the transfer."""

    def __init__(self, rank, world, group=None):
        import torch.distributed as dist
        self.dist, self.rank, self.world, self.group = dist, rank, world, group
        self._comm = None

    def _ops(self, slab):
        # the message buffers of a slab never change: build the op list once per slab
        cached = getattr(self, "_op_cache", None)
        if cached is not None and cached[0] is slab:
            return cached[1]
        ops = self._build_ops(slab)
        self._op_cache = (slab, ops)
        return ops

    def _build_ops(self, slab):
        # only the part of a message buffer that can hold records right now travels
        # (slab.msg_active <= msg_capacity, agreed by all ranks in trim_messages)
        dist = self.dist
        nbytes = message_bytes(getattr(slab, "msg_active", slab.msg_capacity))
        ops = []
        left, right = self.rank - 1, self.rank + 1
        if left >= 0:
            ops.append(dist.P2POp(dist.isend, slab.send_left[:nbytes], left, self.group))
            ops.append(dist.P2POp(dist.irecv, slab.recv_left[:nbytes], left, self.group))
        if right < self.world:
            ops.append(dist.P2POp(dist.isend, slab.send_right[:nbytes], right, self.group))
            ops.append(dist.P2POp(dist.irecv, slab.recv_right[:nbytes], right, self.group))
        return ops

    def forget(self):
        """The slab or its active message size changed: rebuild the op list."""
        self._op_cache = None

    def _run(self, ops):
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()          # with RCCL: the current stream waits, the host does not

    def exchange(self, slab):
        stream = getattr(slab, "stream", None)
        if stream is not None:
            import torch
            with torch.cuda.stream(stream):   # RCCL work is ordered against the slab's stream
                self._run(self._ops(slab))
        else:
            self._run(self._ops(slab))

    def comm_stream(self, slab):
        """The communication stream (high priority: its short border work should not queue behind
        the interior's workgroups); None for a CPU stand-in."""
        if getattr(slab, "stream", None) is None:
            return None
        if self._comm is None:
            import torch
            with torch.cuda.device(slab.device):
                self._comm = torch.cuda.Stream(priority=-1)
                self._arrived = torch.cuda.Event()
        return self._comm

    def begin(self, slab):
        """Start the exchange of the messages packed on the communication stream."""
        comm = self.comm_stream(slab)
        if comm is None:                      # CPU stand-in: nothing to overlap with
            self._run(self._ops(slab))
            return
        import torch
        with torch.cuda.stream(comm):
            self._run(self._ops(slab))
            self._arrived.record(comm)

    def finish(self, slab):
        """The slab's stream waits for the exchange started by begin()."""
        stream = getattr(slab, "stream", None)
        if stream is not None and self._comm is not None:
            stream.wait_event(self._arrived)


class HostStagedTransport(DistTransport):
    """Same exchange with the messages staged through host memory (for process groups without
    device-to-device P2P, e.g. gloo; used to rehearse the multi-rank path on one GPU).
    Synchronous: begin() does the whole exchange, finish() nothing."""

    def comm_stream(self, slab):
        return None          # everything on the slab's stream

    def begin(self, slab):
        self.exchange(slab)

    def finish(self, slab):
        pass

    def exchange(self, slab):
        import torch
        dist = self.dist
        left, right = self.rank - 1, self.rank + 1
        nbytes = message_bytes(getattr(slab, "msg_active", slab.msg_capacity))
        with torch.cuda.stream(slab.stream):
            sends = {k: (getattr(slab, "send_" + k)[:nbytes].cpu()
                         if getattr(slab, "send_" + k) is not None else None)
                     for k in ("left", "right")}
        slab.stream.synchronize()
        recvs = {k: (torch.empty_like(v) if v is not None else None) for k, v in sends.items()}
        ops = []
        if left >= 0:
            ops.append(dist.P2POp(dist.isend, sends["left"], left, self.group))
            ops.append(dist.P2POp(dist.irecv, recvs["left"], left, self.group))
        if right < self.world:
            ops.append(dist.P2POp(dist.isend, sends["right"], right, self.group))
            ops.append(dist.P2POp(dist.irecv, recvs["right"], right, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        with torch.cuda.stream(slab.stream):
            for k in ("left", "right"):
                if recvs[k] is not None:
                    getattr(slab, "recv_" + k)[:nbytes].copy_(recvs[k], non_blocking=False)


class DistSlabStepper:
    """The per-rank loop body.

    overlap=False: pack -> exchange -> unpack -> step, all on one stream.
    overlap=True (default): after one such exchange has delivered the first ghosts,
        step_begin (messages of the planes next to the neighbours are ready early)
        -> exchange on the communication stream  ||  step_end (interior acceleration, integrate)
        -> unpack.
    Both leave the same state behind a step, except that the overlapped loop has already
    exchanged the ghosts for the next one."""

    CHECK_EVERY = 16     # steps between two looks at the slab's error bits (no synchronisation)

    def __init__(self, slab, transport, overlap=True, make_slab=None, cuts=None,
                 rebalance_every=0, imbalance=1.1, trim_every=0, control_group=None):
        """make_slab(cuts, rank, plane_histogram) -> a new, empty slab for planes
        [cuts[rank], cuts[rank + 1]): needed for rebalance() (with `cuts`, the current ones).
        rebalance_every / trim_every: steps between collective re-evaluations of the cut planes
        (when the fullest slab holds more than `imbalance` x the mean) and of the message size
        (0 = never; both synchronise the ranks, so hundreds of steps apart).
        control_group: process group for everything that is NOT the halo exchange - the agreements
        about message size and cut planes and the rows that change owner - with host tensors (a
        gloo group: these must keep working where the device-to-device path does not; default:
        the transport's own group, tensors where that group wants them)."""
        self.slab, self.transport = slab, transport
        self.control_group = control_group
        self.overlap = overlap and hasattr(slab, "step_begin")
        self.make_slab, self.cuts = make_slab, (list(cuts) if cuts is not None else None)
        self.rebalance_every, self.imbalance, self.trim_every = rebalance_every, imbalance, trim_every
        self.rebalances = 0
        self.device_rebalances = 0      # ... of them without the host in the data path
        self.message_growths = 0
        self._trimmed = False           # trim_messages() has been used: all ranks poll from then on
        self._primed = False
        self._steps = 0

    def run(self, steps):
        for _ in range(steps):
            self.step()

    # ---- collectives between steps (every rank calls them at the same step) --------------------
    def _group(self):
        return self.control_group if self.control_group is not None else self.transport.group

    def _device(self):
        """where the control group wants its tensors (RCCL: the slab's GPU; gloo: host)"""
        dist = self.transport.dist
        backend = dist.get_backend(self._group())
        return self.slab.device if backend == "nccl" and hasattr(self.slab, "device") else "cpu"

    def trim_messages(self, slack=1.25, extra=1024):
        """Only the used part of a halo message needs to cross the link: the buffers are sized
        generously once (slab_capacities), but the records actually packed are far fewer - 1.4 MB
        of 3.2 MB in the 4M-particle benchmark.  Every rank looks at the messages it packed last,
        the largest count (x slack + extra records of head room) becomes the size every message
        is packed for, sent and received with from now on; a message that outgrows it raises the
        overflow bit, which stops the run (poll_errors) - call this again then, or periodically
        (trim_every).  Collective; synchronises."""
        import torch
        dist, slab = self.transport.dist, self.slab
        if not hasattr(slab, "send_counts"):
            return slab.msg_capacity
        self._trimmed = True
        want = min(slab.msg_capacity, int(max(slab.send_counts()) * slack) + extra)
        t = torch.tensor([want], dtype=torch.int64, device=self._device())
        if self.transport.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._group())
        slab.msg_active = int(t.item())
        self.transport.forget()
        return slab.msg_active

    def grow_messages_if_needed(self, fill=0.8):
        """A trimmed message must grow BEFORE it overflows (an overflow drops records: the run is
        lost).  Called every CHECK_EVERY steps while the messages are trimmed: every rank looks at
        the record counts an asynchronous copy brought since the last call (no synchronisation
        with the device); when any rank's message is more than `fill` full, all of them go back to
        the size the buffers were allocated for (the next trim_messages() tightens it again).
        Collective over the control group (a few integers)."""
        import torch
        dist, slab = self.transport.dist, self.slab
        if not hasattr(slab, "poll_send_counts") or self.transport.world == 1:
            return False
        if getattr(slab, "msg_active", slab.msg_capacity) >= slab.msg_capacity and not self._trimmed:
            return False
        seen = slab.poll_send_counts()
        wish = 1 if (seen is not None and max(seen) > fill * slab.msg_active) else 0
        t = torch.tensor([wish, -int(slab.msg_capacity)], dtype=torch.int64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._group())
        if int(t[0].item()) == 0:
            return False
        slab.msg_active = min(slab.msg_capacity, -int(t[1].item()))   # what every rank's buffers hold
        self.transport.forget()
        self.message_growths += 1
        return True

    def rebalance(self, force=False):
        """Re-evaluate the cut planes from the current distribution of the particles along z and
        move the particles whose plane changed owner (cuts "re-evaluated
        periodically").  The state itself moves unchanged - ids, positions, velocities, masses - so
        the run continues bit for bit as if the cuts had always been there.  Collective and
        host-staged (download, point-to-point exchange of the rows that change owner, upload into a
        new slab): meant for every few hundred steps.  Returns True if the cuts changed."""
        import torch
        dist, tr, slab = self.transport.dist, self.transport, self.slab
        rank, world, group = tr.rank, tr.world, self._group()
        if world == 1 or self.make_slab is None or self.cuts is None:
            return False
        dev = self._device()
        owned = torch.tensor([slab.status()["owned"]], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(owned) for _ in range(world)]
        dist.all_gather(every, owned, group=group)
        counts = np.array([int(t.item()) for t in every], np.float64)
        if not force and counts.max() <= self.imbalance * counts.mean():
            return False
        if hasattr(slab, "export_records"):
            return self._rebalance_on_device()
        d = slab.download()
        mass = slab.download_mass()
        pos3 = d["pos"].reshape(-1, 3)
        planes = plane_of(slab.params, pos3[:, 2])
        nz = slab.params.full_cells_z
        hist = torch.from_numpy(np.bincount(planes, minlength=nz).astype(np.int64)).to(dev)
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        hist = hist.cpu().numpy()
        new_cuts = cuts_from_histogram(hist, world)
        if new_cuts == self.cuts:
            return False
        # rows {x, y, z, m, vx, vy, vz, id bits} by new owner
        rows = np.empty((mass.size, 8), np.float32)
        rows[:, 0:3] = pos3
        rows[:, 3] = mass
        rows[:, 4:7] = d["vel"].reshape(-1, 3)
        rows[:, 7] = d["ids"].view(np.float32)
        dest = np.searchsorted(np.asarray(new_cuts[1:], np.int64), planes, side="right")
        out = [np.ascontiguousarray(rows[dest == r]) for r in range(world)]
        mine = torch.tensor([o.shape[0] for o in out], dtype=torch.int64, device=dev)
        table = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(table, mine, group=group)          # table[s][r] = rows s sends to r
        ops, inbox, keep_alive = [], {}, []
        for peer in range(world):
            if peer == rank:
                continue
            if out[peer].shape[0]:
                t = torch.from_numpy(out[peer]).to(dev)
                keep_alive.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer, group))
            n_in = int(table[peer][rank].item())
            if n_in:
                inbox[peer] = torch.empty((n_in, 8), dtype=torch.float32, device=dev)
                ops.append(dist.P2POp(dist.irecv, inbox[peer], peer, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if dev != "cpu":
            torch.cuda.synchronize()
        parts = [out[rank]] + [inbox[k].cpu().numpy() for k in sorted(inbox)]
        rows = np.concatenate(parts) if len(parts) > 1 else parts[0]
        rows = rows[np.argsort(rows[:, 7].copy().view(np.uint32), kind="stable")]
        new_slab = self.make_slab(new_cuts, rank, hist)
        new_slab.upload(np.ascontiguousarray(rows[:, 7]).view(np.uint32),
                        np.ascontiguousarray(rows[:, 0:3]).reshape(-1),
                        np.ascontiguousarray(rows[:, 4:7]).reshape(-1),
                        np.ascontiguousarray(rows[:, 3]),
                        all_masses_equal=bool(getattr(slab, "all_masses_equal", False)))
        # the message size the ranks agreed on stays in force (all of them carry it over alike)
        new_slab.msg_active = min(getattr(slab, "msg_active", new_slab.msg_capacity),
                                  new_slab.msg_capacity)
        # ... and so do the timing level, its stride and the pair arithmetic
        if hasattr(slab, "settings") and hasattr(new_slab, "apply_settings"):
            new_slab.apply_settings(slab.settings())
        if hasattr(slab, "close"):
            slab.close()
        self.slab, self.cuts = new_slab, list(new_cuts)
        self._primed = False            # the new slabs have no ghosts yet
        tr.forget()
        self.rebalances += 1
        return True

    def _rebalance_on_device(self):
        """rebalance() without the host in the data path: the owned particles leave the slab as
        message records in device memory (HipSlab.export_records), are sorted into their new
        owners there, the rows that change owner travel point-to-point over the transport's own
        group - device tensors over RCCL; over a gloo group (rehearsals on one GPU) only those rows
        are staged through the host - and the new slab is filled from device records.  The
        agreements (histogram, row counts) stay on the control group."""
        import torch
        dist, tr, slab = self.transport.dist, self.transport, self.slab
        rank, world, group = tr.rank, tr.world, self._group()
        cdev = self._device()
        rec = slab.export_records()                       # float32 [n, 8] on the slab's device
        p = slab.params
        nz = int(p.full_cells_z)
        c = torch.floor(rec[:, 2] * float(np.float32(p.full_cell_inv)))      # plane_of(), on the device
        c = torch.where(torch.isfinite(c), c, torch.full_like(c, -1.0))
        planes = torch.clamp(c, 0, nz - 1).to(torch.int64)
        hist = torch.bincount(planes, minlength=nz).to(torch.int64).to(cdev)
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        hist = hist.cpu().numpy()
        new_cuts = cuts_from_histogram(hist, world)
        if new_cuts == self.cuts:
            return False
        bounds = torch.tensor(new_cuts[1:], dtype=torch.int64, device=rec.device)
        dest = torch.bucketize(planes, bounds, right=True)
        out = [rec[dest == r].contiguous() for r in range(world)]
        mine = torch.tensor([o.shape[0] for o in out], dtype=torch.int64, device=cdev)
        table = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(table, mine, group=group)          # table[s][r] = rows s sends to r
        data_group = tr.group
        on_device = dist.get_backend(data_group) == "nccl"
        ops, inbox, keep_alive = [], {}, []
        for peer in range(world):
            if peer == rank:
                continue
            if out[peer].shape[0]:
                t = out[peer] if on_device else out[peer].cpu()
                keep_alive.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer, data_group))
            n_in = int(table[peer][rank].item())
            if n_in:
                inbox[peer] = torch.empty((n_in, 8), dtype=torch.float32,
                                          device=rec.device if on_device else "cpu")
                ops.append(dist.P2POp(dist.irecv, inbox[peer], peer, data_group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if on_device:
            torch.cuda.synchronize()
        parts = [out[rank]] + [inbox[k].to(rec.device) for k in sorted(inbox)]
        rows = torch.cat(parts) if len(parts) > 1 else parts[0]
        ids = rows[:, 7].contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
        rows = rows[torch.argsort(ids, stable=True)].contiguous()
        new_slab = self.make_slab(new_cuts, rank, hist)
        new_slab.upload_records(rows, bool(getattr(slab, "all_masses_equal", False)))
        new_slab.msg_active = min(getattr(slab, "msg_active", new_slab.msg_capacity), new_slab.msg_capacity)
        new_slab.apply_settings(slab.settings())
        slab.close()
        self.slab, self.cuts = new_slab, list(new_cuts)
        self._primed = False            # the new slabs have no ghosts yet
        tr.forget()
        self.rebalances += 1
        self.device_rebalances += 1
        return True

    def step(self):
        slab, tr = self.slab, self.transport
        # fail loudly: a run that has lost particles (message or capacity overflow, a particle the
        # early exchange missed) stops within 2 * CHECK_EVERY steps instead of running on
        if self._steps % self.CHECK_EVERY == 0:
            if hasattr(slab, "poll_errors"):
                slab.poll_errors()
            if self._trimmed:
                self.grow_messages_if_needed()
        if self._steps > 0:
            if self.rebalance_every and self._steps % self.rebalance_every == 0:
                if self.rebalance():
                    slab = self.slab
            if self.trim_every and self._steps % self.trim_every == 0:
                self.trim_messages()
        self._steps += 1
        if not self.overlap:
            slab.pack()
            tr.exchange(slab)
            slab.unpack()
            slab.step()
            return
        if not self._primed:
            slab.pack()
            tr.exchange(slab)
            slab.unpack()
            self._primed = True
        comm = tr.comm_stream(slab)
        if comm is not None:
            slab.step_begin(comm)
        else:
            slab.step_begin()
        tr.begin(slab)
        slab.step_end()
        tr.finish(slab)
        slab.unpack()


class LocalSlabGroup:
    """All slabs in ONE process (e.g. on one GPU): the messages are handed over by pointer.
    Same kernels and message format as the distributed run; used to check on a single GPU that
    results do not depend on the number of slabs."""

    CHECK_EVERY = 16

    def __init__(self, slabs, overlap=False, exchange_stream=None):
        self._steps = 0
        self.slabs = slabs
        self.overlap = overlap       # the early-exchange protocol (sph_hip_slab_step_begin/end)
        # optional second stream for the border work, as in the distributed run (the messages
        # are then complete in ITS order, and the slabs' stream has to wait for it)
        self.exchange_stream = exchange_stream
        self._primed = False

    def _deliver(self):
        for r, s in enumerate(self.slabs):
            left = self.slabs[r - 1].send_right if r > 0 else None
            right = self.slabs[r + 1].send_left if r + 1 < len(self.slabs) else None
            s.unpack(left, right)

    def step(self):
        if self._steps % self.CHECK_EVERY == 0:
            for s in self.slabs:
                if hasattr(s, "poll_errors"):
                    s.poll_errors()
        self._steps += 1
        if not self.overlap:
            for s in self.slabs:
                s.pack()
            self._deliver()
            for s in self.slabs:
                s.step()
            return
        if not self._primed:
            for s in self.slabs:
                s.pack()
            self._deliver()
            self._primed = True
        for s in self.slabs:
            s.step_begin(self.exchange_stream)
        for s in self.slabs:
            s.step_end()
        if self.exchange_stream is not None:
            self.slabs[0].stream.wait_stream(self.exchange_stream)
        self._deliver()

    def gather(self, n_total):
        """Per-id arrays assembled from every slab's owned particles."""
        out = dict(pos=np.zeros(3 * n_total, np.float32), vel=np.zeros(3 * n_total, np.float32),
                   rho=np.zeros(n_total, np.float32), acc=np.zeros(3 * n_total, np.float32),
                   ncount=np.zeros(n_total, np.int32), owner=np.full(n_total, -1, np.int32))
        for r, s in enumerate(self.slabs):
            d = s.download()
            ids = d["ids"].astype(np.int64)
            assert (out["owner"][ids] == -1).all(), "a particle is owned by two slabs"
            out["owner"][ids] = r
            for k in ("pos", "vel", "acc"):
                out[k].reshape(-1, 3)[ids] = d[k].reshape(-1, 3)
            out["rho"][ids] = d["rho"]
            out["ncount"][ids] = d["ncount"]
        return out
