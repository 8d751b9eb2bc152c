"""Pictures of one temporal layer on different devices (SURVEY.md 8e "Inter (C5)"; BASELINE.json configs[4]: encoder_randomaccess_main10, GOP 8,
8 GPUs).

A picture can be searched as soon as its reference pictures are FINISHED (deblocked, SAO applied, motion compressed: TEncGOP.cpp:1184,1483,1660),
so the pictures of a GOP fall into levels -- level(p) = 1 + max(level of its references) -- and the pictures of one level are independent of
each other: GOP 8 of cfg/encoder_randomaccess_main10.cfg gives {8}, {4}, {2, 6}, {1, 3, 5, 7} after the intra picture.  Rank r of `world`
takes every world-th picture of a level (coding order); after a level every rank holds every finished picture of it: the owners export theirs
as one blob each (hm355_ref_export: border-extended planes + compressed motion field, 28 MB for a 4K picture) and ONE all-gather per level
hands them round (RCCL over xGMI between GPUs: device tensors, no host hop; gloo in the CPU tests), the others import them as device-resident
references (hm355_ref_import).

What the caller has to bring per picture is what TEncGOP hands compressSlice anyway: slice type, QP, lambda, reference lists -- and the
context table (cabac_init_type).  The reference derives that table from the PREVIOUS picture in coding order (determineCabacInitIdx,
TEncSlice.cpp:1083-1093), which chains every picture to its predecessor's bitstream pass; a caller that wants the reference's exact stream
therefore either runs the pictures in coding order on one device, or predicts the table (it changes rarely), verifies it against
`next_cabac_init_type` of the predecessor once that is known and re-encodes on a miss.  The scheduler below takes the tables as given and
reports every predecessor's choice so that the caller can do exactly that (`run_gop` returns them; tests replay the reference's own choices).
The SAO picture-level on / off decision reads the disabled rate of the last picture of the next lower temporal depth in coding order
(TEncSampleAdaptiveOffset.cpp:259-330); that picture lies in an earlier level, so its rates travel with its blob (the four user doubles).
"""
import numpy as np


def levels(pictures):
    """pictures: dicts with "poc" and "refs" (POCs it references), in coding order -> list of levels, each a list of indices into `pictures`
    (coding order inside a level).  A picture whose references are all outside `pictures` (already finished) is level 0."""
    lvl, idx = {}, {p["poc"]: i for i, p in enumerate(pictures)}
    for p in pictures:
        lvl[p["poc"]] = 1 + max([lvl[r] for r in p["refs"] if r in idx and idx[r] < idx[p["poc"]]], default=-1)
    out = [[] for _ in range(1 + max(lvl.values(), default=-1))]
    for i, p in enumerate(pictures):
        out[lvl[p["poc"]]].append(i)
    return out


def owner(position_in_level, world):
    return position_in_level % world


def sao_rate_source(pictures, i):
    """index of the picture whose SAO disabled rates picture i reads (the last one of temporal depth - 1 before it in coding order), or None"""
    d = pictures[i]["depth"]
    for j in range(i - 1, -1, -1):
        if pictures[j]["depth"] == d - 1:
            return j
    return None


def run_gop(engine, pictures, rank, world, all_gather, known=None):
    """Encode `pictures` (coding order) level by level; rank `rank` encodes its share and imports everybody else's finished pictures.
      engine.encode(pic, refs: {poc: handle}, prev_rates) -> (handle of the finished picture as a reference, result, rates (3 floats))
      engine.export(handle, rates) -> blob;  engine.imp(blob) -> (handle, rates);  engine.blob_like() -> an empty blob (for ranks that own
      fewer pictures of a level than others);  all_gather(list of m blobs) -> list over ranks of lists of m blobs
    known: {poc: (handle, rates)} of pictures finished earlier (e.g. the intra picture).  Returns ({poc: result} of this rank's pictures,
    {poc: (handle, rates)} of every picture)."""
    done = dict(known or {})
    results = {}
    for level in levels(pictures):
        mine = [i for k, i in enumerate(level) if owner(k, world) == rank]
        m = (len(level) + world - 1) // world
        blobs = []
        for i in mine:
            p = pictures[i]
            src = sao_rate_source(pictures, i)
            prev = done[pictures[src]["poc"]][1] if src is not None and pictures[src]["poc"] in done else (0.0, 0.0, 0.0)
            handle, res, rates = engine.encode(p, {r: done[r][0] for r in p["refs"]}, prev)
            done[p["poc"]] = (handle, rates)
            results[p["poc"]] = res
            blobs.append(engine.export(handle, rates))
        while len(blobs) < m:
            blobs.append(engine.blob_like())
        if world > 1:
            gathered = all_gather(blobs)
            for k, i in enumerate(level):
                o = owner(k, world)
                if o != rank:
                    done[pictures[i]["poc"]] = engine.imp(gathered[o][k // world])
    return results, done


class TorchAllGather:
    """all_gather of equal-size byte blobs over torch.distributed ("nccl" = RCCL: device tensors; "gloo": host tensors)"""

    def __init__(self, dist, torch, device=None):
        self.dist, self.torch, self.device = dist, torch, device

    def __call__(self, blobs):
        t = [b if hasattr(b, "data_ptr") else self.torch.from_numpy(np.ascontiguousarray(b, np.uint8)) for b in blobs]
        if self.device is not None:
            t = [x.to(self.device) for x in t]
        mine = self.torch.stack(t) if t else self.torch.empty((0, 0), dtype=self.torch.uint8)
        out = [self.torch.empty_like(mine) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, mine)
        return [[o[k] if self.device is not None else o[k].numpy() for k in range(o.shape[0])] for o in out]
