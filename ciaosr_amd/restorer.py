"""Restorers: BasicRestorer / CiaoSR with the reference's call surface.

Mirrors mmedited/models/restorers/basic_restorer.py:35-124 (`__init__`, `forward`, `evaluate`) and
mmedited/models/restorers/ciaosr.py:36-58 (`__init__`), :111-203 (`forward_test`), :218-258
(`clip_test`).  Normalisation, tile blending and de-normalisation run as HIP kernels; tiles are
the unit sharded across GPUs (ciaosr_amd/tile_shard.py).  Training entry points are out of scope.
"""
import math
import numbers
import os.path as osp

import torch
import torch.nn as nn

from . import hip_ops, metrics
from .registry import build_backbone, build_loss


class _Cfg(dict):
    """dict with attribute access (stands in for mmcv.ConfigDict)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _as_cfg(cfg):
    if cfg is None or isinstance(cfg, _Cfg):
        return cfg
    return _Cfg(cfg)


class BasicRestorer(nn.Module):
    allowed_metrics = {'PSNR': metrics.psnr, 'SSIM': metrics.ssim}

    def __init__(self, generator, pixel_loss, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.train_cfg = _as_cfg(train_cfg)
        self.test_cfg = _as_cfg(test_cfg)
        self.fp16_enabled = False
        self.generator = build_backbone(generator)
        # the generator sees the restorer's test_cfg (looked up per call): `allow_f16_substitute` is read from it
        self._bind_generator_cfg()
        self.init_weights(pretrained)
        self.pixel_loss = build_loss(pixel_loss)

    # `test_cfg` is a property so that REASSIGNING it (`restorer.test_cfg = {...}`, what mmedit's apis do) re-binds the generator's view
    # as well; mutating the dict in place is seen without that.
    @property
    def test_cfg(self):
        return self.__dict__.get('_test_cfg')

    @test_cfg.setter
    def test_cfg(self, cfg):
        self.__dict__['_test_cfg'] = _as_cfg(cfg)
        self._bind_generator_cfg()

    def _bind_generator_cfg(self):
        gen = self.__dict__.get('_modules', {}).get('generator', None)
        if gen is not None and hasattr(gen, 'bind_test_cfg'):
            gen.bind_test_cfg(self.__dict__.get('_test_cfg'))

    def init_weights(self, pretrained=None):
        self.generator.init_weights(pretrained)

    def forward(self, lq, gt=None, test_mode=False, **kwargs):
        if test_mode:
            return self.forward_test(lq, gt, **kwargs)
        raise NotImplementedError('forward_train / train_step are out of scope of the MI355X inference path')

    def evaluate(self, output, gt):
        """PSNR/SSIM on uint8 BGR images as basic_restorer.py:101-124."""
        crop_border = self.test_cfg.crop_border
        out_img, gt_img = metrics.tensor2img(output), metrics.tensor2img(gt)
        res = {}
        for metric in self.test_cfg.metrics:
            fn = self.allowed_metrics[metric]
            if 'convert_to' in self.test_cfg:
                res[metric] = fn(out_img, gt_img, crop_border=crop_border, convert_to=self.test_cfg.convert_to)
            else:
                res[metric] = fn(out_img, gt_img, crop_border)
        return res


def tile_starts(n, tile, overlap):
    """ciaosr.py:227-229."""
    stride = tile - overlap
    return list(range(0, n - tile, stride)) + [n - tile]


def tile_grid(h, w, tile, overlap):
    """Row-major (h outer, w inner) list of tile origins, the reference's blend order (ciaosr.py:233-234)."""
    tile = min(tile, h, w)
    return tile, [(hi, wi) for hi in tile_starts(h, tile, overlap) for wi in tile_starts(w, tile, overlap)]


class CiaoSR(BasicRestorer):
    def __init__(self, generator, pixel_loss, rgb_mean=(0.5, 0.5, 0.5), rgb_std=(0.5, 0.5, 0.5), train_cfg=None,
                 test_cfg=None, pretrained=None):
        super().__init__(generator, pixel_loss, train_cfg=train_cfg, test_cfg=test_cfg, pretrained=pretrained)
        self.rgb_mean = tuple(float(v) for v in rgb_mean)
        self.rgb_std = tuple(float(v) for v in rgb_std)
        # host-side float32 copies (the reference keeps them as plain tensors, ciaosr.py:52-58)
        self.lq_mean = torch.FloatTensor(rgb_mean).view(1, -1, 1, 1)
        self.lq_std = torch.FloatTensor(rgb_std).view(1, -1, 1, 1)
        self.gt_mean = torch.FloatTensor(rgb_mean).view(1, 1, -1)
        self.gt_std = torch.FloatTensor(rgb_std).view(1, 1, -1)

    def train_step(self, data_batch, optimizer):
        raise NotImplementedError('training is out of scope of the MI355X inference path')

    # -- pieces of forward_test, separately callable (bench / tile sharding) ---------------------
    @torch.no_grad()
    def normalize(self, lq):
        """(lq - mean) / std  (ciaosr.py:142-144) on the GPU."""
        lq = lq.contiguous().float()
        hip_ops.require_gpu(lq)
        return torch.stack([hip_ops.normalize(lq[b], self.rgb_mean, self.rgb_std) for b in range(lq.shape[0])])

    @torch.no_grad()
    def run_tile(self, x_norm, hi, wi, tile, sf, options=None):
        """One tile of clip_test (ciaosr.py:235-245): [B, th*tw, 3] prediction of the LR crop."""
        patch = x_norm[..., hi:hi + tile, wi:wi + tile].contiguous()
        b = patch.shape[0]
        th, tw = round(patch.shape[-2] * sf), round(patch.shape[-1] * sf)
        coord, cell = hip_ops.make_coord_cell(th, tw, patch.device)       # generated on the GPU, cached per shape
        coord = coord.unsqueeze(0).expand(b, -1, 2)
        cell = cell.unsqueeze(0).expand(b, -1, 2)
        return self.generator(patch, coord, cell, test_mode=True, options=self.options(options)), (th, tw)

    def run_tiles(self, x_norm, origins, tile, sf, options=None):
        """Several equally sized tiles of ONE image (batch 1) through one generator call: the crops are stacked into a batch, so
        the encoder's dense-layer launches are shared (encoder_hip.forward_hwc_batch); the head then runs per tile.  Returns
        ([n, th*tw, 3], (th, tw)); row i is bitwise the run_tile result of origins[i]."""
        patch = torch.cat([x_norm[..., hi:hi + tile, wi:wi + tile] for (hi, wi) in origins], 0).contiguous()
        n = patch.shape[0]
        th, tw = round(patch.shape[-2] * sf), round(patch.shape[-1] * sf)
        coord, cell = hip_ops.make_coord_cell(th, tw, patch.device)
        coord = coord.unsqueeze(0).expand(n, -1, 2)
        cell = cell.unsqueeze(0).expand(n, -1, 2)
        return self.generator(patch, coord, cell, test_mode=True, options=self.options(options)), (th, tw)

    def prepare(self, options=None):
        """Pack every weight the call's precision reads NOW, on the current stream (idempotent; re-packs only what changed) --
        including the 16-bit fragment copies of a 'bf16' / 'f16' call, which are otherwise packed lazily on first use.
        The tile loop runs tiles on several streams: nothing a tile reads may be first built on another tile's stream."""
        opt = self.options(options)
        gen = self.generator
        if hasattr(gen, 'effective_options'):
            opt = gen.effective_options(opt)
        head = getattr(gen, '_head', None)
        enc = getattr(gen, '_encoder_hip', None)
        if enc is not None and enc.supported():
            enc.struct(opt.half)
        if getattr(gen, 'non_local_attn', False):
            gen.cs_attn.packed()
        if head is not None:      # last: the 'bf16-single' form runs a pack-time calibration through the (packed) fp32 trunk and cs_attn
            head.struct(opt.half, single=bool(opt.bf16_single))

    @torch.no_grad()
    def clip_test(self, img_lq, model=None, tile_fn=None, options=None):
        """Tiled inference of one large image (ciaosr.py:218-258).  Returns [B, h*sf*w*sf, 3].

        Tiles are independent, so with `test_cfg.tile_streams = 2` (an extension; default 1) consecutive tiles run on two
        HIP streams: the ramp, first-load and drain phases of one tile's ~450 launches (8.5 us per dense layer, GEMM
        tails, the HBM-bound softmax / patch kernels) fill with the other tile's workgroups instead of idling the chip
        (-2 % fp32, -3.5 % bf16 on a 6-tile image).  Every tile is computed exactly as on one stream (own scratch per
        stream) and the blend stays on the caller's stream in the reference order (h outer, w inner), so the result is
        bitwise the single-stream result.  `tile_streams` is off by default because per-kernel event timings (bench.py's
        roofline leg, rocprof) are meaningless while two streams share the chip.

        `test_cfg.encoder_ahead` (default TRUE since round 5, see `_clip_test_encoder_ahead`) DOES put a second stream under an
        image of more than `tile_batch` tiles: the next batch's trunk runs on a cached side stream (fork / join by events,
        `record_stream` on the hand-over buffers, two batches of feature maps live).  Bitwise the one-stream image; set
        `test_cfg.encoder_ahead = False` for per-kernel timing, rocprof attribution of a multi-batch image, or when calling under
        your own stream capture ("test_cfg extensions")."""
        sf = self.test_cfg.get('scale', None)
        b, c, h, w = img_lq.shape
        tile, origins = tile_grid(h, w, self.test_cfg.get('tile', None), self.test_cfg.get('tile_overlap', None))
        E = torch.zeros(b, c, h * sf, w * sf, dtype=torch.float32, device=img_lq.device)
        Wt = torch.zeros_like(E)
        n_streams = int(self.test_cfg.get('tile_streams', 1) or 1)
        n_batch = self.tile_batch(options)
        if (tile_fn is None and n_streams <= 1 and n_batch > 1 and b == 1 and len(origins) > 1 and img_lq.is_cuda and
                hasattr(getattr(self.generator, '_encoder_hip', None), 'forward_hwc_batch')):
            # `test_cfg.tile_batch` (an extension; default 7 or 8, see tile_batch()) consecutive tiles share the encoder's dense-layer launches; every tile
            # is bitwise the one-at-a-time result and the blend order is the reference's
            if self.test_cfg.get('encoder_ahead', True) and len(origins) > n_batch and getattr(self.generator, '_head', None) is not None:
                return self._clip_test_encoder_ahead(img_lq, tile, origins, n_batch, sf, E, Wt, options)
            for i0 in range(0, len(origins), n_batch):
                group = origins[i0:i0 + n_batch]
                outs, (th, tw) = self.run_tiles(img_lq, group, tile, sf, options)
                for (hi, wi), out in zip(group, outs):
                    hip_ops.tile_blend(E[0], Wt[0], out.contiguous(), hi * sf, wi * sf, th, tw)
            return torch.stack([hip_ops.tile_finalize(E[0], Wt[0])])
        if tile_fn is not None or n_streams <= 1 or len(origins) < 2 or not img_lq.is_cuda:
            for (hi, wi) in origins:
                out, (th, tw) = self.run_tile(img_lq, hi, wi, tile, sf, options) if tile_fn is None else tile_fn(hi, wi)
                for bi in range(b):
                    hip_ops.tile_blend(E[bi], Wt[bi], out[bi].contiguous(), hi * sf, wi * sf, th, tw)
            return torch.stack([hip_ops.tile_finalize(E[bi], Wt[bi]) for bi in range(b)])
        cur = torch.cuda.current_stream(img_lq.device)
        self.prepare(options)
        th = tw = round(tile * sf)
        hip_ops.make_coord_cell(th, tw, img_lq.device)              # cached coordinates exist before any side stream reads them
        streams = self._tile_streams(n_streams, img_lq.device)
        for st in streams:
            st.wait_stream(cur)                                      # the normalised image, E / Wt and the packed weights are ready
        pending = []                                                 # (origin, out, event) in tile order

        def blend_one():
            (hi, wi), out, ev = pending.pop(0)
            cur.wait_event(ev)
            for bi in range(b):
                hip_ops.tile_blend(E[bi], Wt[bi], out[bi], hi * sf, wi * sf, th, tw)

        for i, (hi, wi) in enumerate(origins):
            st = streams[i % n_streams]
            with torch.cuda.stream(st):
                out, _ = self.run_tile(img_lq, hi, wi, tile, sf, options)
                out = out.contiguous()
                ev = torch.cuda.Event()
                ev.record(st)
            out.record_stream(cur)                                   # consumed by the blend on the caller's stream
            pending.append(((hi, wi), out, ev))
            if len(pending) > n_streams:                             # keep at most one finished tile per stream waiting
                blend_one()
        while pending:
            blend_one()
        for st in streams:
            cur.wait_stream(st)
        return torch.stack([hip_ops.tile_finalize(E[bi], Wt[bi]) for bi in range(b)])

    def _clip_test_encoder_ahead(self, img_lq, tile, origins, n_batch, sf, E, Wt, options):
        """`test_cfg.encoder_ahead` (an extension; ON by default since round 5 for images of more than one tile batch -- a product default is
        not chosen for the profiler's convenience: bench.py times its step with it and takes its per-kernel event timings in a separate
        pass with `encoder_ahead = False`, where no second stream shares the chip): the RDN trunk of tile batch k + 1 runs on a side stream
        UNDER the heads of batch k (cs_attn + fused head kernels, the caller's stream).  The trunk's 130 strictly dependent launches per batch
        and the heads' long MFMA kernels fill each other's ramp / drain / memory phases.  Same kernels on the same data in the same
        per-tile order: the image is bitwise the default path's."""
        gen = self.generator
        opt = self.options(options)
        gen._require_hip_trunk(img_lq)                       # an uncovered trunk raises CiaoSRHipError here, not a C-level argument error
        enc = gen._encoder_hip
        dev = img_lq.device
        cur = torch.cuda.current_stream(dev)
        self.prepare(opt)
        th = tw = round(tile * sf)
        coord, cell = hip_ops.make_coord_cell(th, tw, dev)
        side = self._tile_streams(1, dev)[0]
        side.wait_stream(cur)
        groups = [origins[i0:i0 + n_batch] for i0 in range(0, len(origins), n_batch)]

        def trunk(group):
            with torch.cuda.stream(side):
                patches = torch.cat([img_lq[..., hi:hi + tile, wi:wi + tile] for (hi, wi) in group], 0).contiguous().float()
                feats = enc.forward_hwc_batch(patches, opt)
                ev = torch.cuda.Event()
                ev.record(side)
            return patches, feats, ev

        nxt = trunk(groups[0])
        for k, group in enumerate(groups):
            patches, feats, ev = nxt
            if k + 1 < len(groups):
                nxt = trunk(groups[k + 1])                       # queued behind batch k's trunk, runs under batch k's heads
            cur.wait_event(ev)
            patches.record_stream(cur)
            feats.record_stream(cur)
            for j, (hi, wi) in enumerate(group):
                out = gen._head.forward(None, patches[j], coord, cell, gen.eval_bsize, feature_hwc=feats[j], options=opt)
                hip_ops.tile_blend(E[0], Wt[0], out.contiguous(), hi * sf, wi * sf, th, tw)
        cur.wait_stream(side)
        return torch.stack([hip_ops.tile_finalize(E[0], Wt[0])])

    def _tile_streams(self, n, device):
        key = (n, device.index)
        cache = self.__dict__.setdefault('_tile_stream_cache', {})
        if key not in cache:
            cache[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
        return cache[key]

    def tile_batch(self, options=None):
        """Tiles per encoder call (`test_cfg.tile_batch`, an extension; at most 16: 32-bit buffer offsets into the batched block buffer).
        Default 7 where the trunk's dense layers run a kernel whose workgroup covers 16 x 32 pixels -- the F(4x4, 3x3) Winograd kernel
        of the fp32 trunk (dense_direct = 0) and, since round 6, the 16-bit dense kernel (dense_direct != 1): a 192 x 192 tile is 72
        workgroups / items, 8 tiles are 576 = 2.25 rounds of the 256 CUs (a third round at a quarter of the chip), 7 tiles are 504 =
        1.97 -- else 8."""
        v = self.test_cfg.get('tile_batch', None)
        if v is None:
            opt = self.options(options)
            fp32_trunk = opt.precision == 'fp32' or (opt.precision == 'f16' and opt.f16_pairs == 2)       # 'f16x3' keeps the fp32 trunk
            v = 7 if (opt.dense_direct == 0 if fp32_trunk else opt.dense_direct != 1) else 8
        return min(int(v or 1), 16)

    def options(self, options=None):
        """The hip_ops.Options a call runs with: the explicit argument if given, else `test_cfg.precision`
        ('fp32' default | 'bf16'; an extension absent from the reference) + `test_cfg.hip_options` (dict of
        ciaosr_options_t fields).  Nothing process-global: two restorers in one process can differ."""
        if options is not None:
            return hip_ops.as_options(options)
        cfg = self.test_cfg or {}
        extra = dict(cfg.get('hip_options', None) or {})
        prec = cfg.get('precision', None)
        if prec is None and not extra:
            return hip_ops.DEFAULT_OPTIONS
        return hip_ops.Options(prec or 'fp32', **extra)          # 'fp32' | 'bf16' | 'f16' | 'f16-pairs'

    @torch.no_grad()
    def restore(self, lq, coord=None, cell=None, options=None):
        """forward_test body from normalised LR on device to de-normalised, clamped output
        [B,3,round(h*s),round(w*s)] on device (ciaosr.py:142-169) -- the timed region of bench.py.
        `options` / `test_cfg.precision = 'bf16'` (an extension, absent from the reference) runs the dense layers
        with bf16 MFMA inputs; the default is the exact-fp32 path."""
        return self._restore(lq, coord, cell, self.options(options))

    @torch.no_grad()
    def clip_test_any_scale(self, img_lq, ht, wt, options=None):
        """Opt-in tiled inference for non-integer / > 4 scales (tile_plan.py): the reference's LR tiling and
        uniform blending, HR rectangles by pixel-centre membership, tile-local coordinates and cells.
        Returns [B, ht*wt, 3]."""
        from . import tile_plan
        b, c, h, w = img_lq.shape
        tiles = tile_plan.plan(h, w, ht, wt, self.test_cfg.get('tile'), self.test_cfg.get('tile_overlap', 0) or 0)
        E = torch.zeros(b, c, ht, wt, dtype=torch.float32, device=img_lq.device)
        Wt = torch.zeros_like(E)
        for t in tiles:
            patch = img_lq[..., t['y0']:t['y0'] + t['th'], t['x0']:t['x0'] + t['tw']].contiguous()
            coord = t['coord'].to(img_lq.device).unsqueeze(0).expand(b, -1, 2)
            cell = t['cell'].to(img_lq.device).unsqueeze(0).expand(b, -1, 2)
            out = self.generator(patch, coord, cell, test_mode=True, options=self.options(options))
            for bi in range(b):
                hip_ops.tile_blend(E[bi], Wt[bi], out[bi].contiguous(), t['i0'], t['j0'], t['i1'] - t['i0'], t['j1'] - t['j0'])
        return torch.stack([hip_ops.tile_finalize(E[bi], Wt[bi]) for bi in range(b)])

    def _restore(self, lq, coord=None, cell=None, options=None):
        x = self.normalize(lq)
        if self.test_cfg.get('tile', None) and self.test_cfg.get('tile_any_scale', False) and coord is not None:
            ih, iw = lq.shape[-2:]
            s = math.sqrt(coord.shape[1] / (ih * iw))                 # the reference's own size rule (ciaosr.py:166-169)
            pred = self.clip_test_any_scale(x, round(ih * s), round(iw * s), options)
            n_q = pred.shape[1]
        elif self.test_cfg.get('tile', None):
            pred = self.clip_test(x, self.generator, options=options)
            n_q = pred.shape[1]
        else:
            pred = self.generator(x, coord, cell, test_mode=True, options=options)
            n_q = coord.shape[1]
        ih, iw = lq.shape[-2:]
        s = math.sqrt(n_q / (ih * iw))
        H, W = round(ih * s), round(iw * s)
        return torch.stack([hip_ops.denorm_clamp(pred[b].contiguous(), H, W, self.rgb_mean, self.rgb_std)
                            for b in range(lq.shape[0])])

    def graphed_restore(self, lq, coord=None, cell=None, warmup=2, options=None):
        """Capture `restore(lq)` (a few hundred short launches for a 48x48 tile) into one hipGraph and return a
        callable `run(new_lq=None) -> output` that replays it; input and output live in static buffers.  All
        kernels are launched on torch's current stream, workspaces are allocated during the warm-up calls, and
        nothing in the path synchronises, so the whole step is capturable."""
        static_lq = lq.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.prepare(options)
            for _ in range(max(warmup, 1)):
                self.restore(static_lq, coord, cell, options)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # capture on the SAME stream the warm-up ran on: hip_ops' scratch is per stream, so every workspace the captured launches
        # point into already exists (nothing is allocated under capture) ...
        with torch.cuda.graph(graph, stream=side):
            static_out = self.restore(static_lq, coord, cell, options)

        # ... and is handed over to this closure: the graph holds raw pointers into the scratch buffers, the coordinate tensors
        # and the packed weights that were live during capture, so exactly those objects live as long as the graph does and no
        # longer (take_workspaces removes them from hip_ops' cache: a later eager call on a stream that re-uses the handle gets
        # fresh scratch, and dropping `run` frees the memory).  Weights must not change after capture (a repack would be
        # invisible to the captured launches).
        # With test_cfg.tile_streams > 1 or test_cfg.encoder_ahead the captured launches also point into the scratch of the restorer's
        # cached side streams: those buffers are taken too (a later, larger eager call on such a stream would otherwise re-grow -- i.e.
        # free -- a buffer the graph still reads and writes).
        scratch = hip_ops.take_workspaces(side)
        for streams in self.__dict__.get('_tile_stream_cache', {}).values():
            for st in streams:
                scratch.update(hip_ops.take_workspaces(st))
        keep = (scratch, dict(hip_ops._coord_cache),
                [getattr(m, '_packed', None) for m in self.modules()],
                [(getattr(o, '_st', None), getattr(o, '_keep', None), getattr(o, '_mask_keep', None),
                  dict(getattr(o, '_st_half', None) or {}))
                 for m in self.modules() for o in (getattr(m, '_head', None), getattr(m, '_encoder_hip', None)) if o is not None])

        def run(new_lq=None):
            if new_lq is not None:
                static_lq.copy_(new_lq)
            graph.replay()
            return static_out
        run.graph = graph
        run.keep = keep
        return run

    def forward_test(self, lq, gt, coord=None, cell=None, meta=None, save_image=False, save_path=None,
                     iteration=None):
        """Same contract as ciaosr.py:111-203."""
        pred = self.restore(lq, coord, cell)
        if gt is not None:
            shape = [lq.shape[0], pred.shape[2], pred.shape[3], 3]
            gt = gt.view(*shape).permute(0, 3, 1, 2).contiguous()
        if self.test_cfg is not None and self.test_cfg.get('metrics', None):
            assert gt is not None, 'evaluation with metrics must have gt images.'
            results = dict(eval_result=self.evaluate(pred, gt))
        else:
            results = dict(lq=lq.cpu(), output=pred.cpu())
            if gt is not None:
                results['gt'] = gt.cpu()
        if save_image:
            if 'gt_path' in meta[0]:
                folder_name = osp.splitext(osp.basename(meta[0]['gt_path']))[0]
            else:
                folder_name = osp.splitext(osp.basename(meta[0]['lq_path']))[0]
            if isinstance(iteration, numbers.Number):
                save_path = osp.join(save_path, folder_name, f'{folder_name}-{iteration + 1:06d}.png')
            elif iteration is None:
                save_path = osp.join(save_path, f'{folder_name}.png')
            else:
                raise ValueError(f'iteration should be number or None, but got {type(iteration)}')
            from .imageio import imwrite
            imwrite(metrics.tensor2img(pred), save_path)
        return results

    def init_weights(self, pretrained=None, strict=True):
        self.generator.init_weights(pretrained, strict)
