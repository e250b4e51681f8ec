"""TEST INFRASTRUCTURE: a CPU stand-in for smoothed_particle_hydrodynamics_amd.slab.HipSlab.

Same methods (upload / pack / unpack / step / step_begin / step_end / download) and the same message format
(8 int32 header words + 32-byte records), with the step computed by the oracle.  It lets the
slab orchestration (planning, message flow, torch.distributed transport) run on CPU tensors
under gloo.  It lives under tests/ because it calls the oracle; the product never imports it.
"""
import numpy as np
import torch

from smoothed_particle_hydrodynamics_amd.slab import HALO, message_bytes, plane_of


class FakeSlab:
    def __init__(self, oracle, oparams, plane_lo, plane_hi, msg_capacity, has_left, has_right):
        self.o, self.p = oracle, oparams
        self.plane_lo, self.plane_hi = plane_lo, plane_hi
        self.msg_capacity = self.msg_active = msg_capacity
        self.params = oparams
        nbytes = message_bytes(msg_capacity)
        mk = lambda on: torch.zeros(nbytes, dtype=torch.uint8) if on else None
        self.send_left, self.send_right = mk(has_left), mk(has_right)
        self.recv_left, self.recv_right = mk(has_left), mk(has_right)
        self.owned = None      # dict ids,pos(n,3),vel(n,3),mass
        self.ghosts = None
        self.last = None
        self.errors = 0
        self._settings = {}
        self._counts_seen = None

    def upload(self, ids, pos, vel, mass, all_masses_equal):
        self.all_masses_equal = bool(all_masses_equal)
        self.owned = dict(ids=np.asarray(ids, np.uint32).copy(),
                          pos=np.asarray(pos, np.float32).reshape(-1, 3).copy(),
                          vel=np.asarray(vel, np.float32).reshape(-1, 3).copy(),
                          mass=np.asarray(mass, np.float32).copy())
        self.ghosts = None

    @staticmethod
    def _write(msg, sel, part, cap):
        buf = msg.numpy()
        n = int(sel.sum())
        assert n <= cap, "message overflow in the fake slab"
        buf[:32].view(np.int32)[:] = 0
        buf[:32].view(np.int32)[0] = n
        rec = buf[32:32 + 32 * n].view(np.float32).reshape(n, 8)
        rec[:, 0:3] = part["pos"][sel]
        rec[:, 3] = part["mass"][sel]
        rec[:, 4:7] = part["vel"][sel]
        rec[:, 7] = part["ids"][sel].view(np.float32)

    def pack(self):
        o = self.owned
        pl = plane_of(self.p, o["pos"][:, 2])
        if self.send_left is not None:
            self._write(self.send_left, pl < self.plane_lo + HALO, o, self.msg_active)
        if self.send_right is not None:
            self._write(self.send_right, pl >= self.plane_hi - HALO, o, self.msg_active)
        keep = (pl >= self.plane_lo) & (pl < self.plane_hi)
        if self.send_left is None:
            assert (pl >= self.plane_lo).all()
        if self.send_right is None:
            assert (pl < self.plane_hi).all()
        # a migrant that is still inside this slab's halo stays as a ghost for one step: its new
        # owner cannot send it back before the next exchange
        stay = ~keep & (pl >= self.plane_lo - HALO) & (pl < self.plane_hi + HALO)
        self.kept = {k: v[stay] for k, v in o.items()}
        self.owned = {k: v[keep] for k, v in o.items()}
        self.ghosts = None

    @staticmethod
    def _read(msg):
        buf = msg.numpy()
        n = int(buf[:32].view(np.int32)[0])
        rec = buf[32:32 + 32 * n].view(np.float32).reshape(n, 8)
        return dict(ids=rec[:, 7].copy().view(np.uint32), pos=rec[:, 0:3].copy(),
                    vel=rec[:, 4:7].copy(), mass=rec[:, 3].copy())

    def unpack(self, recv_left=None, recv_right=None):
        left = recv_left if recv_left is not None else self.recv_left
        right = recv_right if recv_right is not None else self.recv_right
        parts = [self._read(m) for m in (left, right) if m is not None]
        if not parts:
            self.ghosts = self.kept
            return
        inc = {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}
        pl = plane_of(self.p, inc["pos"][:, 2])
        mine = (pl >= self.plane_lo) & (pl < self.plane_hi)
        halo_ok = (pl >= self.plane_lo - HALO) & (pl < self.plane_hi + HALO)
        if not halo_ok.all():
            self.errors |= 1
        self.owned = {k: np.concatenate([self.owned[k], inc[k][mine]]) for k in self.owned}
        g = ~mine & halo_ok
        self.ghosts = {k: np.concatenate([v[g], self.kept[k]]) for k, v in inc.items()}

    def step(self):
        parts = [self.owned] + ([self.ghosts] if self.ghosts is not None else [])
        allp = {k: np.concatenate([q[k] for q in parts]) for k in self.owned}
        n_own = self.owned["ids"].size
        is_owned = np.zeros(allp["ids"].size, bool)
        is_owned[:n_own] = True
        # the oracle's canonical order inside a cell is the array index: present everything by id
        order = np.argsort(allp["ids"], kind="stable")
        pos = np.ascontiguousarray(allp["pos"][order]).reshape(-1)
        vel = np.ascontiguousarray(allp["vel"][order]).reshape(-1)
        mass = np.ascontiguousarray(allp["mass"][order])
        out = self.o.step(self.p, pos, vel, mass, mode="full")
        sel = is_owned[order]
        self.owned = dict(ids=allp["ids"][order][sel], pos=pos.reshape(-1, 3)[sel],
                          vel=vel.reshape(-1, 3)[sel], mass=mass[sel])
        # what download() returns: the particles this slab owned during the step (as the HIP slab,
        # whose owned range is that of the step's cell build)
        self.last = dict(ids=self.owned["ids"].copy(), pos=self.owned["pos"].copy(),
                         vel=self.owned["vel"].copy(), mass=self.owned["mass"].copy(),
                         rho=out["rho"][sel],
                         acc=out["acc"].reshape(-1, 3)[sel], ncount=out["ncount"][sel])
        self.ghosts = None

    def step_begin(self):
        """Early-exchange protocol: the messages are ready before the step has finished.  Here
        the whole step simply runs first; what matters is the order of calls the stepper makes."""
        self.step()
        self.pack()

    def step_end(self):
        pass

    def download(self):
        l = self.last
        return dict(ids=l["ids"], pos=l["pos"].reshape(-1), vel=l["vel"].reshape(-1),
                    rho=l["rho"], acc=l["acc"].reshape(-1), ncount=l["ncount"])

    def download_mass(self):
        return self.last["mass"]

    def send_counts(self):
        return tuple(int(m.numpy()[:4].view(np.int32)[0]) if m is not None else 0
                     for m in (self.send_left, self.send_right))

    def poll_send_counts(self):
        """as HipSlab.poll_send_counts: what the previous call's request brought"""
        seen, self._counts_seen = self._counts_seen, self.send_counts()
        return seen

    # what DistSlabStepper.rebalance carries over to the slab that replaces this one
    def set_timing(self, level):
        self._settings["timing"] = int(level)

    def set_timing_stride(self, every):
        self._settings["timing_stride"] = int(every)

    def set_arithmetic(self, arithmetic):
        self._settings["arithmetic"] = int(arithmetic)

    def settings(self):
        return dict(self._settings)

    def apply_settings(self, settings):
        self._settings.update(settings)

    def close(self):
        self.closed = True

    def status(self):
        return dict(live=0, owned=int(self.owned["ids"].size), errors=self.errors)
