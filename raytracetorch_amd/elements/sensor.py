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
        self._locs, self._inten, self._ids = [], [], []
        self._dense = []  # (hit planes [3,N], w [N], mask [N] uint8, id [N]) per visit, not yet compacted

    def _record_dense(self, hit_planes, w, mask, ids):
        self._dense.append((hit_planes, w, mask, ids))

    def _flush(self):
        for hit, w, mask, ids in self._dense:
            m = mask.bool()
            if not bool(m.any()):
                continue  # the reference skips a visit nobody hits (scene/sequential.py:25-26)
            self._locs.append(hit[:, m].t())
            self._inten.append(w[m])
            self._ids.append(ids[m])
        self._dense = []

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

    def getSpotSizeParallel_xy(self, query_ids, target_xy=None, norm_ord=2):
        """Per-id weighted moments sum_i w |x - c|^p / (2 sum w) (elements/sensor.py:87-176).
        Returns (spot [K], intensity_sum [K]) in the order of `query_ids`."""
        xyz, w, ids = self.getHitsTensors()
        q = torch.as_tensor(query_ids, device=ids.device, dtype=torch.int32)
        keep = torch.isin(ids, q)
        xy, w, ids = xyz[keep][:, :2], w[keep], ids[keep]
        q_sorted, order = torch.sort(q)
        g = torch.searchsorted(q_sorted, ids.to(torch.int32))
        k = q.shape[0]
        wsum = torch.zeros(k, device=ids.device, dtype=w.dtype).scatter_add(0, g, w)
        denom = torch.where(wsum == 0, torch.ones_like(wsum), wsum)
        g2 = g.unsqueeze(1).expand(-1, 2)
        if target_xy is None:
            cen = torch.zeros(k, 2, device=ids.device, dtype=xy.dtype).scatter_add(0, g2, xy * w[:, None])
            cen = cen / denom[:, None]
        else:
            cen = target_xy[order]
        mom = torch.zeros(k, 2, device=ids.device, dtype=xy.dtype).scatter_add(
            0, g2, w[:, None] * (xy - cen[g]).abs() ** norm_ord)
        spot_sorted = mom.sum(dim=1) / (2 * denom)
        spot = torch.empty_like(spot_sorted)
        spot[order] = spot_sorted
        return spot, wsum
