"""Batched multi-object pose estimator -- the consumer of the hot path.

Same interface as the reference's m3vision plugin
(/root/reference/auto_pose/m3_interface/ae_pose_estimator.py:16-232:
``AePoseEstimator(test_config_path).process(bboxes, color_img, camK, ...)``), but
instead of one ``session.run`` per detection (ae_pose_estimator.py:143-170) all
detections of an image are handled together:

  1. every crop is cut and bilinearly resized on the GPU in one launch
     (extract_square_patch(black_borders=True) + cv2.resize(INTER_LINEAR));
  2. the crops of each object class go through that object's encoder + codebook
     scan as ONE batch;
  3. the translation / rotation-correction geometry of auto_pose6d runs on the
     host per detection (float64, as in the reference).

The results are returned in detection order, one ``PoseEstimate`` per accepted box.
"""
from __future__ import annotations

import ast
import configparser
import os

import numpy as np

from . import ae_factory as factory
from . import session as S
from . import utils as u


class PoseEstimate(object):
    """m3_interface/m3_interfaces.py:57-85."""

    def __init__(self, name='SLC', trafo=np.identity(4), quality=1.0):
        self.name = name
        self.trafo = trafo
        self.quality = quality


class BoundingBox(object):
    """Detection in relative image coordinates with per-class scores
    (m3_interface/m3_interfaces.py BoundingBox: xmin, xmax, ymin, ymax, classes)."""

    def __init__(self, xmin=0., xmax=0., ymin=0., ymax=0., classes=None):
        self.xmin, self.xmax, self.ymin, self.ymax = xmin, xmax, ymin, ymax
        self.classes = classes if classes is not None else {}


class _FrameStage(object):
    """Grow-only pinned-host and device buffers of one estimator on one GPU: the frame (or the union rectangle of its boxes), the
    box rows, the crops, the per-crop results.  process() then allocates nothing per frame and every copy is asynchronous on the
    caller's stream: host -> pinned is a memcpy, pinned -> device and device -> pinned are queued behind / in front of the kernels."""

    def __init__(self, device):
        self.device = device
        self.direct_rows = 8                # frames with up to this many boxes are read by the crop kernel in place (no H2D copy)
        self.img_host = self.img_dev = None
        self.crops = self.z = self.score = self.idx_host = None
        self.multi_ws = None                # scratch of the one-call-per-frame path (aae_detect_nn_multi)

    @staticmethod
    def _grow(torch, old, n, dtype, device=None, pin=False):
        if old is not None and old.numel() >= n:
            return old
        n = max(int(n), 1)
        return torch.empty((n,), dtype=dtype, pin_memory=True) if pin else torch.empty((n,), dtype=dtype, device=device)

    def upload_frame_and_rows(self, torch, frame, rows):
        """frame uint8 [h,w,c] (any strides) and the int32 [total,5] box rows behind it in ONE pinned staging buffer and ONE
        asynchronous copy (two copies cost two trips through the copy engine: ~8 us each in front of a 82 us query).
        Returns the device views (image [h,w,c], rows [total,5])."""
        h, w, c = frame.shape
        n = h * w * c
        off = (n + 15) // 16 * 16
        total = off + rows.nbytes
        self.img_host = self._grow(torch, self.img_host, total, torch.uint8, pin=True)
        self.img_dev = self._grow(torch, self.img_dev, total, torch.uint8, self.device)
        host = self.img_host.numpy()
        dst = host[:n].reshape(h, w, c)
        # The crop kernel reads nothing but the pixels INSIDE the boxes (crop_resize_u8.h: a tap outside its box is black).  When the
        # boxes cover less than half of the rectangle -- a handful of detections spread over a 1080p frame -- only their rectangles
        # are copied into the staging image, at their own positions (8 boxes: 0.9 MB instead of 3.7-6 MB of host memcpy in front of the
        # first launch); what lies between them is stale and never read.
        sparse = False
        if 2 <= len(rows) <= 64 and h * w >= 65536:        # (one box: the rectangle IS the box; tiny frames: the bookkeeping costs more than the copy)
            x0 = np.clip(rows[:, 0], 0, w)
            y0 = np.clip(rows[:, 1], 0, h)
            x1 = np.clip(rows[:, 0] + rows[:, 2], 0, w)
            y1 = np.clip(rows[:, 1] + rows[:, 3], 0, h)
            sparse = 2 * int(((x1 - x0) * (y1 - y0)).sum()) < h * w
        if sparse:
            for a, b, cc, d in zip(y0.tolist(), y1.tolist(), x0.tolist(), x1.tolist()):
                if b > a and d > cc:
                    dst[a:b, cc:d] = frame[a:b, cc:d]
        else:
            np.copyto(dst, frame)
        host[off:total].view(np.int32)[:] = rows.reshape(-1)
        if len(rows) <= self.direct_rows:
            # a few boxes: the crop kernel reads its pixels and the rows straight out of the pinned buffer (device-accessible) --
            # it touches the boxes only, not the rectangle around them, and a trip through the copy engine costs ~10 us in front of
            # the query (1 detection 188 -> 172 us, 4: 345 -> 338, 8: level; 16 and more: the copy wins, 64: 2.78 vs 2.82 ms)
            return self.img_host[:n].view(h, w, c), self.img_host[off:total].view(torch.int32).view(-1, 5)
        self.img_dev[:total].copy_(self.img_host[:total], non_blocking=True)
        return self.img_dev[:n].view(h, w, c), self.img_dev[off:total].view(torch.int32).view(-1, 5)

    def reserve(self, torch, total, latent, crop_shape):
        """room for `total` detections of this frame: crops [total,h,w,c], latents, scores (device), indices (pinned host: the
        scan writes them there itself)"""
        per_crop = int(np.prod(crop_shape))
        self.crops = self._grow(torch, self.crops, total * per_crop, torch.uint8, self.device)
        self.z = self._grow(torch, self.z, total * latent, torch.float32, self.device)
        self.score = self._grow(torch, self.score, total, torch.float32, self.device)
        self.idx_host = self._grow(torch, self.idx_host, total, torch.int64, pin=True)


class AePoseEstimator(object):

    def __init__(self, test_config_path=None, codebooks=None, train_args=None, upright=False, topk=1, camPose=False, share_workspaces=True):
        """Either ``test_config_path`` (the m3 cfg with an [auto_pose] section, as in the
        reference) or explicit ``codebooks`` / ``train_args`` dicts keyed by class name."""
        self._process_requirements = ['color_img', 'camK', 'bboxes']
        self.all_codebooks, self.all_train_args, self.pad_factors, self.patch_sizes = {}, {}, {}, {}
        self._image_format = {'color_format': 'bgr', 'color_data_type': np.uint8, 'depth_data_type': np.float32}
        self.sess = S.Session()
        if test_config_path is not None:
            test_args = configparser.ConfigParser(inline_comment_prefixes="#")
            if not test_args.read(test_config_path):
                raise FileNotFoundError('test config not found: %s' % test_config_path)
            workspace_path = os.environ.get('AE_WORKSPACE_PATH')
            if workspace_path is None:
                raise RuntimeError('Please define a workspace path: export AE_WORKSPACE_PATH=/path/to/workspace')
            # what the estimator asks its caller to deliver (ae_pose_estimator.py:41-43 reads the three keys and eval()s
            # the type names; here a type name is looked up among the NumPy scalar types instead of being evaluated)
            fmt = self._image_format
            fmt['color_format'] = test_args.get('auto_pose', 'color_format', fallback=fmt['color_format'])
            for key in ('color_data_type', 'depth_data_type'):
                if test_args.has_option('auto_pose', key):
                    fmt[key] = self._numpy_type(test_args.get('auto_pose', key))
            camPose = test_args.getboolean('auto_pose', 'camPose')
            upright = test_args.getboolean('auto_pose', 'upright')
            topk = test_args.getint('auto_pose', 'topk')
            self.class_2_encoder = ast.literal_eval(test_args.get('auto_pose', 'class_2_encoder'))
            for clas_name, experiment in self.class_2_encoder.items():
                full_name = experiment.split('/')
                experiment_name = full_name.pop()
                experiment_group = full_name.pop() if len(full_name) > 0 else ''
                log_dir = u.get_log_dir(workspace_path, experiment_name, experiment_group)
                ckpt_dir = u.get_checkpoint_dir(log_dir)
                targs = configparser.ConfigParser(inline_comment_prefixes="#")
                targs.read(u.get_train_config_exp_file_path(log_dir, experiment_name))
                self._register(clas_name, factory.build_codebook_from_name(experiment_name, experiment_group), targs)
                factory.restore_checkpoint(self.sess, S.Saver(scope=experiment_name), ckpt_dir)
        else:
            if not codebooks or not train_args:
                raise ValueError('pass a test config path, or codebooks= and train_args= dicts keyed by class name')
            self.class_2_encoder = {k: k for k in codebooks}
            for k in codebooks:
                self._register(k, codebooks[k], train_args[k])
        if topk > 1:
            raise NotImplementedError('topk > 1 not implemented (as in the reference, ae_pose_estimator.py:36-39)')
        # The objects of one estimator run one after the other on ONE stream: by default they scratch in the same device memory (one
        # encoder workspace -- 973 MB at batch 256 -- instead of one per object).  That rebinds the .ws of engines the caller may
        # own: share_workspaces=False leaves them alone; close() / unshare_workspaces() puts the engines' own buffers back.  While
        # shared, an engine used from another stream raises instead of scribbling over a running call's scratch (engine._Workspace).
        self._shared_engines = []
        if share_workspaces:
            try:
                from .engine import share_workspaces as _share
                encs = [getattr(getattr(c, '_encoder', None), 'engine', None) for c in self.all_codebooks.values()]
                cbs = [getattr(c, 'engine', None) for c in self.all_codebooks.values()]
                self._shared_engines = _share(encs) + _share(cbs)
            except ImportError:
                pass
        self._camPose, self._upright, self._topk = bool(camPose), bool(upright), int(topk)
        self.upload_union_only = True      # process(): upload the union rectangle of the boxes instead of the frame
        self.poll_results = True           # process(): watch the pinned index buffer instead of waiting on an event per chunk (exact fp32 only)
        self.multi_call = True             # process(): a frame with several classes is ONE C call (aae_detect_nn_multi: one launch per layer across the classes with <= 4 detections)
        self.geometry_chunk = 16           # process(): classes with more than 2 x this many detections go to the GPU in chunks (_chunk_sizes), the last of this size
        if self._camPose:
            self._process_requirements.append('camPose')

    def _chunk_plan(self, counts):
        """How the detections of a frame go to the GPU: ``counts`` = detections per class in launch order -> chunk sizes per
        class.  A chunk's float64 geometry (~10 us per detection) runs on the host while the GPU works on everything queued
        behind it (~33 us per crop), so a chunk may be three times as large as what follows it -- large chunks cost the
        encoder less per crop (31 us at B = 48 against 35 at 16) -- and only the geometry of the very last chunk stays
        exposed: it is geometry_chunk detections, or the whole last class when that has at most twice as many.  Chunks never
        span classes (one encoder per object).  64 of one class -> 48 + 16; 256 -> 192 + 48 + 16; 43 + 21 of two classes ->
        one chunk each."""
        c = max(1, int(self.geometry_chunk))          # (<= 0 would never shrink `left`)
        plan = [[] for _ in counts]
        behind = 0
        for ci in range(len(counts) - 1, -1, -1):
            left, sizes = int(counts[ci]), []
            while left > 0:
                if behind < c:                                # (nothing, or less than a chunk, behind it: the end of the frame)
                    take = left if left <= 2 * c else c
                else:
                    take = left if left <= 3 * behind + c else 3 * behind     # (no sliver in front: a remainder of at most one chunk joins its neighbour)
                take = max(int(take), 1)
                sizes.append(take)
                left -= take
                behind += take
            plan[ci] = sizes[::-1]
        return plan

    def _chunk_sizes(self, n):
        """_chunk_plan for the detections of a single class"""
        return self._chunk_plan([n])[0]

    def _register(self, clas_name, codebook, targs):
        self.all_codebooks[clas_name] = codebook
        self.all_train_args[clas_name] = targs
        self.pad_factors[clas_name] = targs.getfloat('Dataset', 'PAD_FACTOR')
        self.patch_sizes[clas_name] = (targs.getint('Dataset', 'W'), targs.getint('Dataset', 'H'))

    def close(self):
        """Free the device state of every object (N encoders + N codebooks live in one process,
        ae_pose_estimator.py:61-78); the estimator must not be used afterwards."""
        self.unshare_workspaces()
        for cb in self.all_codebooks.values():
            if hasattr(cb, 'close'):
                cb.close(close_encoder=True)
        self.all_codebooks, self.all_train_args = {}, {}

    def unshare_workspaces(self):
        """give every engine its own scratch buffer back (engines that outlive the estimator, or move to other streams)"""
        from .engine import unshare_workspaces as _unshare
        _unshare(self._shared_engines)
        self._shared_engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def set_parameter(self, string_name, string_val):
        pass

    def query_process_requirements(self):
        return self._process_requirements

    @staticmethod
    def _numpy_type(text):
        """'np.uint8' / 'numpy.float32' / 'uint8' -> the NumPy scalar type."""
        name = text.strip().split('.')[-1]
        t = getattr(np, name, None)
        if not (isinstance(t, type) and issubclass(t, np.generic)):
            raise ValueError('[auto_pose] data type %r is not a NumPy scalar type' % text)
        return t

    def query_image_format(self):
        return dict(self._image_format)

    # ------------------------------------------------------------------ crops
    @staticmethod
    def box_rows(boxes_xywh, pad_factor):
        """[x, y, w, h, size] int32 rows exactly as extract_square_patch derives them
        (ae_pose_estimator.py:108-109): astype(int32) truncation, size = int(max(h, w) * pad)."""
        rows = np.empty((len(boxes_xywh), 5), dtype=np.int32)
        if len(boxes_xywh):
            xywh = np.array(boxes_xywh, dtype=np.float64).reshape(-1, 4).astype(np.int32)      # float64 -> int32 truncates like the per-box cast
            rows[:, :4] = xywh
            # size: int(np.maximum(h, w) * pad) -- int32 maximum times a Python float = float64 product, truncated
            pad = pad_factor if isinstance(pad_factor, np.ndarray) else float(pad_factor)       # (a factor per row: the estimator's frame-wide call)
            rows[:, 4] = (np.maximum(xywh[:, 3], xywh[:, 2]) * pad).astype(np.int64)
        return rows

    def extract_square_patches(self, scene_img, boxes_xywh, pad_factor, resize=(128, 128)):
        """Batched extract_square_patch(..., INTER_LINEAR, black_borders=True): device uint8
        [D, resize[1], resize[0], C].  ``resize`` is (W, H) as in cv2."""
        from .engine import crop_resize
        return crop_resize(scene_img, self.box_rows(boxes_xywh, pad_factor), (resize[1], resize[0]))

    # ---------------------------------------------------------------- process
    def process(self, bboxes, color_img, camK, depth_img=None, camPose=None, rois3ds=[], mm=False):
        H, W = color_img.shape[:2]
        if isinstance(color_img, np.ndarray) and color_img.dtype != np.uint8:
            color_img = color_img.astype(np.uint8)      # the reference assigns the crop into a uint8 canvas (ae_pose_estimator.py:113-126)
        accepted = []                                   # (detection index, class, box_xywh)
        for j, box in enumerate(bboxes):
            pred_clas = max(box.classes, key=box.classes.get)
            if pred_clas not in self.class_2_encoder:
                print('%s not contained in config class_names %s' % (pred_clas, list(self.class_2_encoder.keys())))
                continue
            box_xywh = [box.xmin * W, box.ymin * H, (box.xmax - box.xmin) * W, (box.ymax - box.ymin) * H]
            if box_xywh[0] < 0 or box_xywh[1] < 0 or box_xywh[2] < 0 or box_xywh[3] < 0:
                print('invalid bb', box_xywh)
                continue
            accepted.append((j, pred_clas, box_xywh))
        if not accepted:
            return []

        # Only the pixels some box covers travel to the device: the black-border crop reads nothing outside
        # its box, so the union rectangle of the (integer) boxes with the boxes shifted into it gives the same
        # crops bit for bit as the whole frame -- a 1080p frame is 6 MB, one detection a few hundred KB.
        off_x = off_y = 0
        frame = color_img
        if self.upload_union_only:
            ints = [np.array(bb).astype(np.int32) for _, _, bb in accepted]
            L, T = int(min(b[0] for b in ints)), int(min(b[1] for b in ints))
            R = min(W, int(max(b[0] + b[2] for b in ints)))
            Bm = min(H, int(max(b[1] + b[3] for b in ints)))
            if L < R and T < Bm and (R - L) * (Bm - T) <= 0.6 * W * H:
                frame = color_img[T:Bm, L:R]
                off_x, off_y = L, T
        classes = sorted(set(c for _, c, _ in accepted))
        first = self.all_codebooks[classes[0]]
        device = getattr(first._encoder.engine, 'device', None)
        if isinstance(frame, np.ndarray) and device is not None and getattr(device, 'type', 'cpu') == 'cuda' and all(
                getattr(self.all_codebooks[c]._encoder.engine, 'device', None) == device for c in classes):
            poses = self._process_staged(accepted, classes, frame, off_x, off_y, camK, device)
        else:
            poses = self._process_plain(accepted, classes, frame, off_x, off_y, camK)
        out = []
        for j in sorted(poses):
            clas, R, t = poses[j]
            H_est = np.eye(4)
            H_est[:3, :3] = R
            H_est[:3, 3] = t if mm else t / 1000.
            if self._camPose:
                H_est = np.dot(camPose, H_est)
            out.append(PoseEstimate(name=clas, trafo=H_est))
        return out

    def _process_plain(self, accepted, classes, frame, off_x, off_y, camK):
        """{detection index: (class, R [3,3], t [3])} through the public Codebook calls (any engine, e.g. the CPU doubles of the tests)"""
        image_dev = None
        poses = {}
        for clas in classes:
            members = [(j, bb) for j, c, bb in accepted if c == clas]
            codebook = self.all_codebooks[clas]
            if image_dev is None:
                import torch
                image_dev = torch.from_numpy(np.ascontiguousarray(frame)).to(codebook._encoder.engine.device)
            crops = self.extract_square_patches(image_dev, [[bb[0] - off_x, bb[1] - off_y, bb[2], bb[3]] for _, bb in members],
                                                self.pad_factors[clas], resize=self.patch_sizes[clas])
            idcs = np.atleast_1d(codebook.nearest_rotation(self.sess, crops, top_n=1, upright=self._upright, return_idcs=True))
            if hasattr(codebook, 'poses_from_indices'):
                Rs, ts = codebook.poses_from_indices(idcs, [bb for _, bb in members], camK, self.all_train_args[clas])
            else:                                                  # (a stand-in with the reference's per-detection method only)
                per = [codebook.pose_from_indices([idx], bb, camK, self.all_train_args[clas]) for (_, bb), idx in zip(members, idcs)]
                Rs, ts = [r.squeeze() for r, _ in per], [t.squeeze() for _, t in per]
            for k, (j, _) in enumerate(members):
                poses[j] = (clas, Rs[k], ts[k])
        return poses

    @staticmethod
    def _pinned_stores_arrive(torch, device):
        """Does a kernel's store to pinned host memory become visible to the host WITHOUT a stream synchronisation?  (It does for
        coherent pinned memory, the default; with HIP_HOST_COHERENT=0 or another allocator it may not.)  Probed once per device: a
        two-element unpack_pairs launch writes into a pinned buffer and the host watches for at most 50 ms."""
        import ctypes
        import time
        from . import _lib
        from .engine import _on_device, _stream_ptr
        try:
            lib = _lib.load()
            dst = torch.full((2,), -1, dtype=torch.int64).pin_memory()
            sc = torch.empty((2,), dtype=torch.float32, device=device)
            src = torch.tensor([[5, 0], [6, 0]], dtype=torch.int64, device=device)
            torch.cuda.current_stream().synchronize()
            with _on_device(device):                     # (the C entry point directly: the Python wrapper insists on device outputs)
                rc = lib.aae_unpack_pairs(ctypes.c_void_p(src.data_ptr()), None, 2, 2, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
                                          _stream_ptr(torch))
            if rc != 0:
                return False
            view = dst.numpy()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.05:
                if view.min() >= 0:
                    return bool(view[0] == 5 and view[1] == 6)
            torch.cuda.current_stream().synchronize()
            return False
        except Exception:
            return False

    @staticmethod
    def _await_indices(torch, chunk, queued):
        """Wait until every index of `chunk` (a view of the pinned result buffer, -1 = not there yet) has been stored by the scan's
        last block.  The spin is bounded by WALL TIME -- 2 ms + 0.2 ms per detection queued up to this chunk, several times what the
        GPU needs -- and then hands over to a blocking stream synchronisation, after which the indices are there or the runtime has
        reported why not.  (Only the indices are awaited: the latents and scores of the chunk in the estimator's device scratch are
        complete once the stream has drained, not necessarily when process() returns.)"""
        import time
        deadline, spins = None, 0
        last = chunk[-1:]                                          # (the scan's last block stores a chunk's indices in order: watch the last one, then check all)
        while last[0] < 0 or chunk.min() < 0:                      # (one 8-byte store per index; every index of the chunk must have landed)
            spins += 1
            if spins & 63:
                continue
            now = time.perf_counter()
            if deadline is None:
                deadline = now + 2e-3 + 2e-4 * queued
            elif now > deadline:
                torch.cuda.current_stream().synchronize()
                if chunk.min() < 0:
                    raise RuntimeError('aae_detect_nn: the indices of a chunk never arrived')
                return

    def _process_staged(self, accepted, classes, frame, off_x, off_y, camK, device):
        """The same on the GPU with everything queued before the first wait: frame (union rectangle) and box rows travel in ONE
        pinned staging buffer and one copy, every class (in chunks when it has many detections) is ONE C call -- aae_detect_nn:
        crop + resize, encoder, top-1 query -- whose indices are written straight into pinned host memory, with an event behind
        it; the float64 geometry of chunk k then runs on the host while the GPU is busy with chunk k + 1.  Nothing is
        allocated per frame.  Same answers as _process_plain: bit for bit with multi_call = False; with multi_call = True the classes of a
        frame share launch plans (aae_encode_nn_multi), so latents differ by fp32 summation order (<= 2.3e-6 of their scale) and the
        index only where the top-2 cosine gap is below that."""
        import torch
        stage = self.__dict__.setdefault('_stages', {}).get(device)
        if stage is None:
            stage = self._stages[device] = _FrameStage(device)
        crop_shapes = set((self.patch_sizes[c][1], self.patch_sizes[c][0]) for c in classes)
        if len(crop_shapes) != 1:
            return self._process_plain(accepted, classes, frame, off_x, off_y, camK)      # (objects trained at different crop sizes)
        latent_sizes = set(int(self.all_codebooks[c]._encoder.latent_space_size) for c in classes)
        if len(latent_sizes) != 1:
            return self._process_plain(accepted, classes, frame, off_x, off_y, camK)      # (objects with different latent sizes: one staging row width does not fit all)
        oh, ow = crop_shapes.pop()
        C = int(frame.shape[2])
        total = len(accepted)
        J = latent_sizes.pop()
        from .engine import _on_device
        with _on_device(device):                      # (torch.cuda.device costs ~8 us per entry even when the device is current already)
            stage.reserve(torch, total, J, (oh, ow, C))
            # box rows of all classes, class by class; classes with many detections are cut into chunks so that the host's
            # float64 geometry of chunk k runs while the GPU works on chunk k + 1 (the geometry of the LAST chunk is all that
            # stays exposed: 16 detections instead of a whole class)
            groups, at = [], 0
            by_class = [[(j, bb) for j, c, bb in accepted if c == clas] for clas in classes]
            plan = self._chunk_plan([len(m) for m in by_class])
            # the [x, y, w, h, size] rows of every class in one go (box_rows with a pad factor per row: the same float64 product, truncated)
            rows_all = self.box_rows([[bb[0] - off_x, bb[1] - off_y, bb[2], bb[3]] for members in by_class for _, bb in members],
                                     np.array([float(self.pad_factors[clas]) for clas, members in zip(classes, by_class) for _ in members], dtype=np.float64))
            for clas, members, sizes in zip(classes, by_class, plan):
                n = len(members)
                a = 0
                for step in sizes:
                    groups.append((clas, members[a:a + step], at + a, step))
                    a += step
                at += n
            image_dev, rows_dev = stage.upload_frame_and_rows(torch, frame, rows_all)
            crops_all = stage.crops[:total * oh * ow * C].view(total, oh, ow, C)
            z_all = stage.z[:total * J].view(total, J)
            events = []
            idx_host = stage.idx_host.numpy()
            # exact fp32 (the default): the host learns that a chunk is done from the indices themselves -- the scan's last block
            # stores them into pinned host memory, the host watches the sentinel it put there disappear.  An event behind the
            # chunk costs a signal + wake-up on top (~8 us per frame with one detection).  Split precision keeps the events: a
            # chunk that left the fp16 range is recomputed on the stream first (settle()).
            if self.poll_results and device not in self.__dict__.setdefault('_poll_probed', {}):
                self._poll_probed[device] = self._pinned_stores_arrive(torch, device)         # once per device
            poll = (self.poll_results and self._poll_probed.get(device, False)
                    and all(self.all_codebooks[c]._encoder.engine.options.get('precision', 0) == 0 for c in classes))
            if poll:
                idx_host[:total] = -1
            exact = all(self.all_codebooks[c]._encoder.engine.options.get('precision', 0) == 0 for c in classes)
            if self.multi_call and exact and len(groups) >= 2:
                # ONE C call per FRAME (aae_detect_nn_multi): the crops of every class in one launch, then the classes with at most four
                # detections -- a frame's boxes spread over the classes, ae_pose_estimator.py:143-170 -- share one launch per layer
                # (six launches per distinct detection count instead of six per class); larger classes take the per-object path inside
                # the same call.  The rows are already in item order (class by class, chunk by chunk).
                from .engine import _Workspace, detect_nn_multi
                if stage.multi_ws is None:
                    stage.multi_ws = _Workspace(device)
                items = []
                for clas, members, a, n in groups:
                    codebook = self.all_codebooks[clas]
                    items.append((codebook._encoder.engine, codebook.engine, n, int(codebook._dataset._kw['num_cyclo']) if self._upright else 1))
                detect_nn_multi(items, image_dev, rows_dev[:total], crops_all, z_all, stage.idx_host[:total], stage.score[:total], stage.multi_ws)
                ev = None
                if not poll:
                    ev = torch.cuda.Event()
                    ev.record()
                events = [ev] * len(groups)
                # the per-object items (more than four detections) are queued first and finish first: their geometry goes first
                order = sorted(range(len(groups)), key=lambda k: (groups[k][3] <= 4, k))
            else:
                order = list(range(len(groups)))
                for clas, members, a, n in groups:
                    codebook = self.all_codebooks[clas]
                    stride = int(codebook._dataset._kw['num_cyclo']) if self._upright else 1
                    # ONE C call per chunk: crop + resize, encoder, top-1 query; the indices land in pinned host memory directly
                    codebook._encoder.engine.detect_nn(codebook.engine, image_dev, rows_dev[a:a + n], n, stride, crops_all[a:a + n], z_all[a:a + n],
                                                       stage.idx_host[a:a + n], stage.score[a:a + n])
                    if poll:
                        events.append(None)
                    else:
                        ev = torch.cuda.Event()
                        ev.record()
                        events.append(ev)
            poses = {}
            # while the GPU works: the part of the float64 geometry that does not depend on the matched rows
            prepared = [self.all_codebooks[clas].poses_prepare([bb for _, bb in members], camK, self.all_train_args[clas])
                        for clas, members, _, _ in groups]
            for k in order:
                (clas, members, a, n), ev, prep = groups[k], events[k], prepared[k]
                codebook = self.all_codebooks[clas]
                if ev is None:
                    self._await_indices(torch, idx_host[a:a + n], a + n)
                else:
                    ev.synchronize()
                    codebook._encoder.engine.settle()              # (split precision, out of range: recomputed in fp32, indices rewritten in place)
                idcs = idx_host[a:a + n].copy()
                Rs, ts = codebook.poses_from_indices(idcs, [bb for _, bb in members], camK, self.all_train_args[clas], prepared=prep)
                for k, (j, _) in enumerate(members):
                    poses[j] = (clas, Rs[k], ts[k])
        return poses
