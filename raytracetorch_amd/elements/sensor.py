"""``Sensor``: a transmitting surface that records where rays cross it
(reference: elements/sensor.py:9-176).

Per recorded hit the reference keeps (hit_local [3], intensity before this surface, ray id).  The
trace kernels write those densely (one slot per ray plus a hit mask); the compacted per-hit lists
``hitLocs`` / ``hitIntensity`` / ``hitID`` are materialised lazily, in ray order per visit, so the
reference's list semantics hold while hot loops that only need moments never compact.
"""
import torch

from .. import program as P
from ..phys import Transmit
from .parent import Element


class Sensor(Element):

    def __init__(self, shape):
        super().__init__()
        self.shape = shape
        self.surface_functions.extend([Transmit()] * len(shape))
        self.reset()

    def _record(self, j: int):
        rec = super()._record(j)
        rec["flags"] |= P.F_SENSOR
        return rec

    # ---- recording --------------------------------------------------------------------------
    def reset(self):
        # plain attributes, set past nn.Module.__setattr__ (which costs microseconds per assignment and is called
        # once per trace by the optimisation goals)
        d = self.__dict__
        d["_locs"], d["_inten"], d["_ids"] = [], [], []
        d["_dense"] = []  # (hit planes [3,N], w [N], mask [N] uint8, id [N]) per visit, not yet compacted
        d["_dense_bytes"] = 0
        d["_pending"] = []  # hit lists of fused bounce loops nobody has waited for yet; always older than `_dense`
        d["_pending_bytes"] = 0  # device memory those launches keep alive (hit lists and their input planes)

    # A dense record holds 17 bytes per *ray*, the reference's list entry 17 bytes per *hit*: a sensor that is neither
    # read nor reset for many visits (a long stepping loop) compacts its pending records once they pass this size.
    _DENSE_LIMIT = 4 << 30
    _PENDING_LIMIT = 64          # entries
    _PENDING_BYTES = 1 << 30     # device memory the pending entries may keep alive

    def _record_dense(self, hit_planes, w, mask, ids):
        self._dense.append((hit_planes, w, mask, ids))
        d = self.__dict__
        d["_dense_bytes"] += 17 * int(w.shape[0])
        if d["_dense_bytes"] > self._DENSE_LIMIT and len(self._dense) > 1:
            self._flush()

    def _record_visit(self, visit):
        """A visit whose tensors are still views to be made (`kernels.SensorVisit`, a 4-sequence like the tuples
        above)."""
        d = self.__dict__
        d["_dense"].append(visit)
        d["_dense_bytes"] += 17 * visit.n
        if d["_dense_bytes"] > self._DENSE_LIMIT and len(d["_dense"]) > 1:
            self._flush()

    def _record_pending(self, entries_of, nbytes=0):
        """The crossings of a launch that is still running: `entries_of(self)` gives this sensor's list entries
        [(locs, w, ids)] once somebody reads them (the first call waits for the device).  `nbytes` = the device memory
        the entry keeps alive until then (the launch's hit list and its input planes)."""
        if self._dense:
            self._flush()   # keeps the order of the entries: what is pending is older than what is dense
        d = self.__dict__
        if d["_pending"] and (len(d["_pending"]) >= self._PENDING_LIMIT or d["_pending_bytes"] + nbytes > self._PENDING_BYTES):
            # a sensor that is neither read nor reset over many launches settles the old entries (which waits for
            # them and frees what they held) before it takes another: the memory held is bounded in bytes, whatever
            # the size of the bundles
            self._resolve_pending()
        d["_pending"].append(entries_of)
        d["_pending_bytes"] += int(nbytes)

    def _resolve_pending(self):
        pending, self.__dict__["_pending"] = self._pending, []
        self.__dict__["_pending_bytes"] = 0
        for entries_of in pending:
            for locs, w, ids in entries_of(self):
                self._locs.append(locs)
                self._inten.append(w)
                self._ids.append(ids)

    def _record_hits(self, locs, w, ids):
        """One list entry of already compacted hits (rows in ray order)."""
        self._flush()
        self._locs.append(locs)
        self._inten.append(w)
        self._ids.append(ids)

    def _flush(self):
        if self._pending:
            self._resolve_pending()
        for hit, w, mask, ids in self._dense:
            m = mask.bool()
            if not bool(m.any()):
                continue  # the reference skips a visit nobody hits (scene/sequential.py:25-26)
            self._locs.append(hit[:, m].t())
            self._inten.append(w[m])
            self._ids.append(ids[m])
        self.__dict__["_dense"] = []
        self.__dict__["_dense_bytes"] = 0

    @property
    def hitLocs(self):
        self._flush()
        return self._locs

    @property
    def hitIntensity(self):
        self._flush()
        return self._inten

    @property
    def hitID(self):
        self._flush()
        return self._ids

    def forward(self, rays, surf_idx):
        """Element.forward plus a record of every given ray (elements/sensor.py:22-39)."""
        o = self._forward_full(rays, surf_idx)
        self._flush()
        self._locs.append(o["hit_local"])
        self._inten.append(rays.intensity)
        self._ids.append(rays.id)
        return o["hit_global"], o["new_dir"], o["imult"]

    # ---- queries ------------------------------------------------------------------------------
    def getHitsTensors(self, ray_id=None):
        """(locs [M,3], intensities [M], ids [M]) over all recorded hits (elements/sensor.py:46-65)."""
        locs = torch.cat(self.hitLocs, dim=0)
        inten = torch.cat(self.hitIntensity, dim=0)
        ids = torch.cat(self.hitID, dim=0)
        if ray_id is not None:
            m = ids == int(ray_id)
            locs, inten, ids = locs[m], inten[m], ids[m]
        return locs, inten, ids

    def getSpotSizeID_xy(self, ray_id, target_xy=None, norm_ord=2):
        """The reference's single-id variant (elements/sensor.py:67-85), kept expression for expression on the
        compacted hit lists: it weights by the intensities of *all* hits and sums the centroid over both coordinates,
        so it only evaluates when those shapes happen to broadcast, and fails as the reference does otherwise."""
        hit_xyz, hit_intensity, hit_id = self.getHitsTensors()
        chosen = hit_id == ray_id
        xyz, inten = hit_xyz[chosen], hit_intensity[chosen]
        total = torch.sum(hit_intensity)
        if target_xy is None:
            target_xy = torch.sum(xyz[:, 0:2] * inten[:, None]) / total
        diff = xyz[:, 0:2] - target_xy[None, :]
        return torch.sum(hit_intensity * torch.abs(diff) ** norm_ord, dim=0) / total

    def _hit_records(self):
        """(x, y, w, mask or None, id) device arrays of every recorded visit, dense or compacted."""
        if self._pending:
            self._resolve_pending()
        recs = [(hit[0], hit[1], w, mask, ids) for hit, w, mask, ids in self._dense]
        for locs, w, ids in zip(self._locs, self._inten, self._ids):
            recs.append((locs[:, 0].contiguous(), locs[:, 1].contiguous(), w.contiguous(), None, ids.contiguous()))
        return recs

    def getSpotSizeParallel_xy(self, query_ids, target_xy=None, norm_ord=2):
        """Per-id weighted moments sum_i w |x - c|^p / (2 sum w) (elements/sensor.py:87-176), as two segmented
        reductions over the recorded hits (``rtt_spot_groups_f32``; nothing is compacted, sorted or gathered).
        Returns (spot [K] in the order of `query_ids`, intensity_sum [K] in the order of the *sorted* ids, as
        the reference does).  The result carries no autograd graph."""
        from .. import kernels as K
        recs = self._hit_records()
        if not recs:
            raise RuntimeError("no hits recorded")   # the reference fails in torch.cat of empty lists
        dev = recs[0][2].device
        q = torch.as_tensor(query_ids, device=dev).to(torch.int64).reshape(-1)
        inside = (q >= -128) & (q <= 127)
        g = torch.where(inside, q + 128, torch.zeros_like(q))
        sums = torch.zeros((256, 3), device=dev, dtype=torch.float64)
        for x, y, w, mask, ids in recs:
            K.spot_groups(x, y, w, mask, ids, 0, None, float(norm_ord), sums)
        pick = lambda t: torch.where(inside[:, None], t[g], torch.zeros_like(t[g]))
        s0 = pick(sums)
        wsum = s0[:, 0]
        denom = torch.where(wsum == 0, torch.ones_like(wsum), wsum)
        cen = s0[:, 1:3] / denom[:, None] if target_xy is None else torch.as_tensor(target_xy, device=dev).to(torch.float64)
        centers = torch.zeros((256, 2), device=dev, dtype=torch.float64)
        centers[g[inside]] = cen[inside]
        mom = torch.zeros((256, 3), device=dev, dtype=torch.float64)
        for x, y, w, mask, ids in recs:
            K.spot_groups(x, y, w, mask, ids, 1, centers, float(norm_ord), mom)
        m1 = pick(mom)
        spot = (m1[:, 0] + m1[:, 1]) / (2 * denom)
        order = torch.sort(q).indices
        return spot.to(torch.float32), wsum[order].to(torch.float32)
