"""Pose post-processing with the reference's call signatures
(pvn3d/lib/utils/pvn3d_eval_utils.py): per-point keypoint votes -> MeanShift clustering ->
least-squares pose.

  cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, obj_id)
      -> [pred_RT]                                   (reference :156-201)
  cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter)
      -> (pred_cls_ids, pred_pose_lst)               (reference :37-110)
  cal_batch_poses_lm / cal_batch_poses               batched, sync-free forms used by bench.py
  eval_metric / eval_metric_lm                       ADD / ADD-S per object (reference :113-136, :204-221)
  eval_one_frame_pose(_lm)                           reference :140-153, :223-236 (same `item` tuples)
  TorchEval.eval_pose_parallel / cal_auc / cal_lm_add  reference :240-387

The reference launches ~12 torch kernels per mean-shift iteration per fit and synchronises
with the host every iteration; here each call is a handful of launches covering every fit of
every object (see _vote_engine.py); ADD / ADD-S of every object of every frame of a batch are
one more launch (csrc/metrics.hip).  Mesh points come from ``Basic_Utils.get_pointxyz_cuda``
(dataset files, or ``set_pointxyz`` -- the datasets are not part of this repository).
"""
import numpy as np
import torch

from . import _vote_engine as _eng
from .basic_utils import Basic_Utils

_bs_utils = Basic_Utils()
RADIUS = 0.08  # hard-coded in the reference (:44, :163)


def _mesh_kps(cls, ds_type, use_ctr):
    kps = _bs_utils.get_kps(cls, ds_type=ds_type)
    if use_ctr:
        kps = np.concatenate((kps, _bs_utils.get_ctr(cls, ds_type=ds_type).reshape(1, 3)), axis=0)
    return torch.from_numpy(kps.astype(np.float32))


def cal_batch_poses_lm(pclds, masks, ctr_ofs, pred_kp_ofs, use_ctr, n_cls, use_ctr_clus_flter,
                       obj_id, poll_every=8, async_limit=None):
    """Batched cal_frame_poses_lm: pclds (F,N,3), masks (F,N), ctr_ofs (F,1,N,3),
    pred_kp_ofs (F,K,N,3).  Returns the engine dict (poses (F,3,4) float64 on the device, ...)."""
    mesh = _mesh_kps(obj_id, "linemod", True)
    return _eng.frames_pose_single_class(pclds, masks, ctr_ofs, pred_kp_ofs, mesh, cls_id=1,
                                         use_ctr=use_ctr, use_ctr_clus_flter=use_ctr_clus_flter,
                                         radius=RADIUS, poll_every=poll_every, async_limit=async_limit)


def cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, obj_id):
    """One LineMOD frame: pcld (N,3), mask (N), ctr_of (1,N,3), pred_kp_of (K,N,3) -> [3x4 ndarray]."""
    res = cal_batch_poses_lm(pcld.unsqueeze(0), mask.unsqueeze(0), ctr_of.unsqueeze(0),
                             pred_kp_of.unsqueeze(0), use_ctr, n_cls, use_ctr_clus_flter, obj_id)
    return [res["poses"][0].cpu().numpy()]


_YCB_MESH = {}


def _ycb_mesh_all(n_cls, use_ctr=True):
    key = (n_cls, use_ctr)
    if key not in _YCB_MESH:
        _YCB_MESH[key] = torch.stack([_mesh_kps(c, "ycb", True) for c in range(1, n_cls)], 0)
    return _YCB_MESH[key]


def cal_batch_poses(pclds, masks, ctr_ofs, pred_kp_ofs, use_ctr, n_cls, use_ctr_clus_flter,
                    poll_every=8, async_limit=None):
    """Batched cal_frame_poses over F frames and class ids 1..n_cls-1 (engine dict)."""
    return _eng.frames_pose_multi_class(pclds, masks, ctr_ofs, pred_kp_ofs, _ycb_mesh_all(n_cls),
                                        n_cls, _bs_utils.ycb_r_lst, use_ctr=use_ctr,
                                        use_ctr_clus_flter=use_ctr_clus_flter, radius=RADIUS,
                                        poll_every=poll_every, async_limit=async_limit)


class GraphedFramePoses(object):
    """HIP-graph replay of the vote -> cluster -> pose call for frames of ONE static shape -- the reference evaluates one
    frame per call (test_mini_batch_size = 1, pvn3d/common.py:41), where the ~100 launches and three host polls of the
    call, not the GPU, set its latency.

        g = GraphedFramePoses("ycb", pcld, mask, ctr_of, pred_kp_of, n_cls=22)        # tensors (1, N, ...): ONE YCB frame ("lm": F frames)
        res = g(pcld, mask, ctr_of, pred_kp_of)                                      # the engine dict, like cal_batch_poses

    The captured sequence enqueues at most `async_limit` MeanShift iterations per fit batch and no host poll; fits that
    have not finished by then mark themselves (iters < 0, summarised in `unfinished_min`).  `__call__` reads that flag together with the results and,
    if any fit is unfinished (heavy-tailed votes), repeats the frame through the ordinary polled call -- same results
    either way.  kind: "lm" (cal_batch_poses_lm, needs obj_id) or "ycb" (cal_batch_poses)."""

    def __init__(self, kind, pclds, masks, ctr_ofs, pred_kp_ofs, n_cls, use_ctr=True, use_ctr_clus_flter=None, obj_id=None,
                 async_limit=8, warmup=2):
        assert kind in ("lm", "ycb") and pclds.is_cuda
        if kind == "ycb" and pclds.dim() == 3 and pclds.size(0) != 1:
            # several YCB frames take their instance list from the masks through a host read (torch.nonzero in
            # frames_pose_multi_class): not capturable, and a replay would keep the capture's list.  One frame instantiates
            # every class slot instead, which is a fixed launch sequence.
            raise ValueError("GraphedFramePoses('ycb', ...) captures ONE frame per call (got %d)" % pclds.size(0))
        self.kind, self.n_cls, self.use_ctr, self.obj_id = kind, n_cls, use_ctr, obj_id
        self.flt = (kind == "ycb") if use_ctr_clus_flter is None else use_ctr_clus_flter
        self.async_limit = async_limit
        self.static = [t.clone() for t in (pclds, masks, ctr_ofs, pred_kp_ofs)]
        self.fallbacks = 0
        cur = torch.cuda.current_stream(pclds.device)
        side = torch.cuda.Stream(device=pclds.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # allocator pools, cached constants, LDS opt-ins
                self._run(async_limit)
        cur.wait_stream(side)
        torch.cuda.synchronize(pclds.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.res = self._run(async_limit)
            self.unfinished = self.res["unfinished_min"]      # < 0: some fit batch of the call hit the limit

    def _run(self, limit, poll_every=8):
        p, m, c, k = self.static
        if self.kind == "lm":
            return cal_batch_poses_lm(p, m, c, k, self.use_ctr, self.n_cls, self.flt, self.obj_id, poll_every=poll_every,
                                      async_limit=limit)
        return cal_batch_poses(p, m, c, k, self.use_ctr, self.n_cls, self.flt, poll_every=poll_every, async_limit=limit)

    def __call__(self, pclds, masks, ctr_ofs, pred_kp_ofs):
        for dst, src in zip(self.static, (pclds, masks, ctr_ofs, pred_kp_ofs)):
            if dst.shape != src.shape:
                raise RuntimeError("GraphedFramePoses was captured for shape %s" % (tuple(dst.shape),))
            dst.copy_(src)
        self.graph.replay()
        if int(self.unfinished.item()) < 0:               # the one host read of the call (results are read next anyway)
            self.fallbacks += 1
            return self._run(None)
        return self.res


def cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter):
    """One YCB frame -> (pred_cls_ids ndarray, [3x4 ndarray per predicted class])."""
    res = cal_batch_poses(pcld.unsqueeze(0), mask.unsqueeze(0), ctr_of.unsqueeze(0),
                          pred_kp_of.unsqueeze(0), use_ctr, n_cls, use_ctr_clus_flter)
    present = res["present"][0].cpu().numpy()
    poses = res["poses"][0].cpu().numpy()
    pred_cls_ids = (np.nonzero(present)[0] + 1).astype(np.int64)
    return pred_cls_ids, [poses[c - 1] for c in pred_cls_ids]


YCB_SYM_CLS_IDS = [13, 16, 19, 20, 21]     # common.py:82
LM_SYM_CLS_IDS = [10, 11]                   # common.py:93


def _lists(n_cls):
    return [list() for _ in range(n_cls)], [list() for _ in range(n_cls)]


def eval_metric(cls_ids, pred_pose_lst, pred_cls_ids, RTs, mask, label, n_cls=22, bs_utils=None):
    """ADD / ADD-S of every ground-truth object of one YCB frame (reference :113-136): a class
    that was not predicted scores against the all-zero pose (:124)."""
    bs_utils = bs_utils or _bs_utils
    cls_add_dis, cls_adds_dis = _lists(n_cls)
    ids, preds, gts, meshes = [], [], [], []
    dev = RTs.device
    pred_cls_ids = np.asarray(pred_cls_ids).reshape(-1)
    for icls, cls_id in enumerate(cls_ids):
        cid = int(cls_id.reshape(-1)[0].item()) if torch.is_tensor(cls_id) else int(np.asarray(cls_id).reshape(-1)[0])
        if cid == 0:
            break
        where = np.where(pred_cls_ids == cid)[0]
        if len(where) == 0:
            pred_RT = torch.zeros(3, 4, device=dev)
        else:
            pred_RT = torch.from_numpy(np.asarray(pred_pose_lst[where[0]]).astype(np.float32)).to(dev)
        ids.append(cid)
        preds.append(pred_RT)
        gts.append(RTs[icls].to(torch.float32))
        meshes.append(bs_utils.get_pointxyz_cuda(cid, ds_type="ycb"))
    if ids:
        add, adds = _eng.add_adds_batch(meshes, torch.stack(preds), torch.stack(gts))
        add, adds = add.cpu().tolist(), adds.cpu().tolist()
        for cid, a, s_ in zip(ids, add, adds):
            cls_add_dis[cid].append(a)
            cls_adds_dis[cid].append(s_)
            cls_add_dis[0].append(a)
            cls_adds_dis[0].append(s_)
    return cls_add_dis, cls_adds_dis


def eval_metric_lm(cls_ids, pred_pose_lst, RTs, mask, label, obj_id, n_cls=22, bs_utils=None):
    """ADD / ADD-S of the single LineMOD object of a frame (reference :204-221)."""
    bs_utils = bs_utils or _bs_utils
    cls_add_dis, cls_adds_dis = _lists(n_cls)
    dev = RTs.device
    pred_RT = torch.from_numpy(np.asarray(pred_pose_lst[0]).astype(np.float32)).to(dev)
    mesh = bs_utils.get_pointxyz_cuda(obj_id, ds_type="linemod")
    add, adds = _eng.add_adds_batch([mesh], pred_RT[None], RTs[0].to(torch.float32)[None])
    a, s_ = float(add[0].item()), float(adds[0].item())
    cls_add_dis[obj_id].append(a)
    cls_adds_dis[obj_id].append(s_)
    cls_add_dis[0].append(a)
    cls_adds_dis[0].append(s_)
    return cls_add_dis, cls_adds_dis


def eval_one_frame_pose(item):
    """Reference :140-153; `item` as zipped by TorchEval.eval_pose_parallel."""
    pcld, mask, ctr_of, pred_kp_of, RTs, cls_ids, use_ctr, n_cls, min_cnt, use_ctr_clus_flter, label, epoch, ibs = item
    pred_cls_ids, pred_pose_lst = cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter)
    return eval_metric(cls_ids, pred_pose_lst, pred_cls_ids, RTs, mask, label, n_cls=n_cls)


def eval_one_frame_pose_lm(item):
    """Reference :223-236."""
    (pcld, mask, ctr_of, pred_kp_of, RTs, cls_ids, use_ctr, n_cls, min_cnt, use_ctr_clus_flter, label, epoch, ibs,
     obj_id) = item
    pred_pose_lst = cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, obj_id)
    return eval_metric_lm(cls_ids, pred_pose_lst, RTs, mask, label, obj_id, n_cls=n_cls)


class TorchEval(object):
    """The reference's TorchEval (:238-392): pose estimation for a batch + ADD / ADD-S bookkeeping."""

    def __init__(self, n_cls=22, bs_utils=None, log_eval_dir=None, verbose=True):
        self.n_cls = n_cls
        self.cls_add_dis = [list() for _ in range(n_cls)]
        self.cls_adds_dis = [list() for _ in range(n_cls)]
        self.cls_add_s_dis = [list() for _ in range(n_cls)]
        self.sym_cls_ids = []
        self.bs_utils = bs_utils or _bs_utils
        self.log_eval_dir = log_eval_dir
        self.verbose = verbose
        self.last_poses = None

    def _say(self, *a):
        if self.verbose:
            print(*a)

    def cal_auc(self):
        """Per-class and overall ADD / ADD-S / ADD(-S) AUC (reference :249-296); returns the dict the
        reference pickles (and pickles it when log_eval_dir is set)."""
        add_auc_lst, adds_auc_lst, add_s_auc_lst = [], [], []
        for cls_id in range(1, self.n_cls):
            self.cls_add_s_dis[cls_id] = self.cls_adds_dis[cls_id] if cls_id in YCB_SYM_CLS_IDS \
                else self.cls_add_dis[cls_id]
            self.cls_add_s_dis[0] += self.cls_add_s_dis[cls_id]
        for i in range(self.n_cls):
            add_auc_lst.append(self.bs_utils.cal_auc(self.cls_add_dis[i]))
            adds_auc_lst.append(self.bs_utils.cal_auc(self.cls_adds_dis[i]))
            add_s_auc_lst.append(self.bs_utils.cal_auc(self.cls_add_s_dis[i]))
            if i == 0:
                continue
            self._say(self.bs_utils.ycb_cls_lst[i - 1] if i - 1 < len(self.bs_utils.ycb_cls_lst) else i)
            self._say("***************add:\t", add_auc_lst[-1])
            self._say("***************adds:\t", adds_auc_lst[-1])
            self._say("***************add(-s):\t", add_s_auc_lst[-1])
        self._say("Average of all object:")
        self._say("***************add:\t", np.mean(add_auc_lst[1:]))
        self._say("***************adds:\t", np.mean(adds_auc_lst[1:]))
        self._say("***************add(-s):\t", np.mean(add_s_auc_lst[1:]))
        self._say("All object (following PoseCNN):")
        self._say("***************add:\t", add_auc_lst[0])
        self._say("***************adds:\t", adds_auc_lst[0])
        self._say("***************add(-s):\t", add_s_auc_lst[0])
        sv_info = dict(add_dis_lst=self.cls_add_dis, adds_dis_lst=self.cls_adds_dis, add_auc_lst=add_auc_lst,
                       adds_auc_lst=adds_auc_lst, add_s_auc_lst=add_s_auc_lst)
        self._dump(sv_info, 'pvn3d_eval_cuda_{}_{}_{}.pkl'.format(adds_auc_lst[0], add_auc_lst[0], add_s_auc_lst[0]))
        return sv_info

    def cal_lm_add(self, obj_id, test_occ=False, diameter_m=None):
        """LineMOD summary for one object (reference :298-343).  diameter_m: the object's diameter in
        metres; default = `lm_r_lst[obj_id]['diameter'] / 1000` of the reference's models_info.yml
        (bundled, tools/import_obj_kps.py), as the reference looks it up at :314."""
        cls_id = obj_id
        if diameter_m is None:
            from ... import synth as _synth
            z = _synth.obj_kps()
            ids = list(z["lm_diameter_ids"])
            if int(obj_id) in ids:
                diameter_m = float(z["lm_diameter_mm"][ids.index(int(obj_id))]) / 1000.0
        self.cls_add_s_dis[cls_id] = self.cls_adds_dis[cls_id] if obj_id in LM_SYM_CLS_IDS \
            else self.cls_add_dis[cls_id]
        self.cls_add_s_dis[0] += self.cls_add_s_dis[cls_id]
        add_auc = self.bs_utils.cal_auc(self.cls_add_dis[cls_id])
        adds_auc = self.bs_utils.cal_auc(self.cls_adds_dis[cls_id])
        add_s_auc = self.bs_utils.cal_auc(self.cls_add_s_dis[cls_id])
        sv_info = dict(add_dis_lst=self.cls_add_dis, adds_dis_lst=self.cls_adds_dis, add_auc_lst=[add_auc],
                       adds_auc_lst=[adds_auc], add_s_auc_lst=[add_s_auc])
        self._say("***************add auc:\t", add_auc)
        self._say("***************adds auc:\t", adds_auc)
        self._say("***************add(-s) auc:\t", add_s_auc)
        if diameter_m is not None:
            d = diameter_m * 0.1
            sv_info["add"] = np.mean(np.array(self.cls_add_dis[cls_id]) < d) * 100
            sv_info["adds"] = np.mean(np.array(self.cls_adds_dis[cls_id]) < d) * 100
            self._say("***************add < 0.1 diameter:\t", sv_info["add"])
            self._say("***************adds < 0.1 diameter:\t", sv_info["adds"])
        self._dump(sv_info, 'pvn3d_eval_cuda_{}_{}.pkl'.format(obj_id, "occlusion" if test_occ else ""))
        return sv_info

    def _dump(self, info, name):
        if self.log_eval_dir:
            import os
            import pickle as pkl
            with open(os.path.join(self.log_eval_dir, name), "wb") as f:
                pkl.dump(info, f)

    def merge_lst(self, targ, src):
        for i in range(len(targ)):
            targ[i] += src[i]
        return targ

    def eval_pose_parallel(self, pclds, rgbs, masks, pred_ctr_ofs, gt_ctr_ofs, labels, cnt,
                           cls_ids, RTs, pred_kp_ofs, min_cnt=20, merge_clus=False, bbox=False,
                           ds='YCB', cls_type=None, use_p2d=False, vote_type=None,
                           use_ctr_clus_flter=True, use_ctr=True, ds_type="ycb", obj_id=0):
        """Same arguments as the reference (:345-351).  The reference fans the frames of a batch
        out to a thread pool (:373-380); here the poses of the whole batch are one device-side
        pipeline and, when ground-truth poses `RTs` and mesh points are available, ADD / ADD-S of
        every object of every frame are one more launch, merged into ``cls_add_dis`` /
        ``cls_adds_dis`` exactly as the reference does.  Returns the per-frame pose lists (also in
        ``self.last_poses``)."""
        masks = masks.long()
        bs = pclds.size(0)
        if ds_type == "ycb":
            res = cal_batch_poses(pclds, masks, pred_ctr_ofs, pred_kp_ofs, use_ctr, self.n_cls,
                                  use_ctr_clus_flter)
            present = res["present"].cpu().numpy()
            poses = res["poses"].cpu().numpy()
            out = []
            for f in range(poses.shape[0]):
                ids = np.nonzero(present[f])[0] + 1
                out.append((ids, [poses[f, c - 1] for c in ids]))
        else:
            res = cal_batch_poses_lm(pclds, masks, pred_ctr_ofs, pred_kp_ofs, use_ctr, self.n_cls,
                                     use_ctr_clus_flter, obj_id)
            poses = res["poses"].cpu().numpy()
            out = [[poses[f]] for f in range(poses.shape[0])]
        self.last_poses = out
        if RTs is not None and cls_ids is not None:
            # metrics of every frame first, merged only when all of them succeeded; missing mesh points are an
            # error (an evaluation that silently accumulates nothing reports AUC 0) unless
            # `self.poses_only_without_mesh` is set
            per_frame = []
            try:
                for f in range(bs):
                    if ds_type == "ycb":
                        per_frame.append(eval_metric(cls_ids[f].long(), out[f][1], out[f][0], RTs[f], masks[f],
                                                     labels[f] if labels is not None else None, n_cls=self.n_cls,
                                                     bs_utils=self.bs_utils))
                    else:
                        per_frame.append(eval_metric_lm(cls_ids[f].long(), out[f], RTs[f], masks[f],
                                                        labels[f] if labels is not None else None, obj_id,
                                                        n_cls=self.n_cls, bs_utils=self.bs_utils))
            except FileNotFoundError as e:
                if not getattr(self, "poses_only_without_mesh", False):
                    raise RuntimeError("ground-truth poses were given but the object's mesh points are not available "
                                       "(%s); provide the dataset files, call bs_utils.set_pointxyz(), or set "
                                       "TorchEval.poses_only_without_mesh = True to skip ADD / ADD-S" % e)
                per_frame = []
            for add_l, adds_l in per_frame:
                self.cls_add_dis = self.merge_lst(self.cls_add_dis, add_l)
                self.cls_adds_dis = self.merge_lst(self.cls_adds_dis, adds_l)
        return out
