"""Counterpart of the reference trainer base class, `src/eoe/training/ad_trainer.py:93-662`, for the hot path:
`train_cls` (:356-471) and `eval_cls` (:473-550) inner loops, the class x seed `run` loop (:177-354) with
`weight_reset` copies (:31-34,232-243), NaN retry (:257-280), snapshot loading (:552-615), and the three
abstract objective hooks (:624-662) with unchanged signatures.

What differs, deliberately (SURVEY.md sections 3.1 and 8):
  * the optimiser is eoe_amd.FusedAdam (one kernel per step) -- constructed here exactly where the reference
    constructs torch.optim.Adam (:383), same hyper-parameters, driven by the stock MultiStepLR (:384);
  * anomaly scores and the loss stay on the device during an epoch; they are copied to the host once per epoch
    for the NaN check and the AUC (:447-455) instead of forcing two host syncs per step (:436,442);
  * the per-half masked Normalize (:413-425) is one affine with the normal class's statistics, fused into the
    encoder's first kernel when the encoder supports it (SURVEY.md section 8a note 1);
  * optional data parallelism (one process per GPU): each rank trains on its rows of every step batch,
    loss = sum(local) / global batch, gradients summed over RCCL, scores/labels all-gathered for the AUC.
Datasets, loggers with tensorboard/PDF output, MSMs and the CLIP text objective are out of scope.
"""
import json
import os
from abc import ABC, abstractmethod
from copy import deepcopy
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .. import ops, parallel
from ..metrics import ROC, PRC, roc_auc, average_precision, auc_ap_device
from ..optim import FusedAdam


class NanGradientsError(RuntimeError):
    pass


def weight_reset(m: torch.nn.Module):
    # ad_trainer.py:31-34
    reset_parameters = getattr(m, "reset_parameters", None)
    if callable(reset_parameters):
        m.reset_parameters()


class JsonLogger:
    """minimal stand-in for `src/eoe/utils/logger.py` (out of scope): JSON lines + snapshots with the reference's
    snapshot dict layout {net, opt, sched, epoch, ds_statistics} (`logger.py:318-340`)"""

    def __init__(self, logdir: Optional[str] = None, active: bool = True):
        # under data parallelism every rank runs the trainer; only rank 0 writes files (the ranks' weights are identical, and
        # for BatchNorm encoders rank 0's running statistics are the ones kept)
        rank0 = int(os.environ.get("RANK", "0")) == 0
        self.dir, self.active = logdir, active and logdir is not None and rank0
        if self.active:
            os.makedirs(os.path.join(logdir, "snapshots"), exist_ok=True)

    def print(self, msg: str):
        print(msg, flush=True)

    warning = logtxt = print

    def logjson(self, name: str, obj):
        if self.active:
            with open(os.path.join(self.dir, name + ".json"), "w") as f:
                json.dump(obj, f, default=lambda o: None if (isinstance(o, float) and o != o) else o)

    def add_scalar(self, *a, **k):
        pass

    def snapshot(self, name: str, net: torch.nn.Module, opt=None, sched=None, epoch: int = None, **kwargs):
        if not self.active:
            return None
        path = os.path.join(self.dir, "snapshots", name + ".pt")
        data = {"net": net.state_dict(), "opt": opt.state_dict() if opt is not None else None,
                "sched": sched.state_dict() if sched is not None else None, "epoch": epoch}
        data.update(kwargs)
        torch.save(data, path)
        return path


class ADTrainer(ABC):
    AD_MODES = ("one_vs_rest", "leave_one_out", "fifty_fifty")
    KEEP_SNAPSHOT_IN_RAM = False

    def __init__(self, model: torch.nn.Module, train_transform=None, test_transform=None, dataset=None,
                 oe_dataset=None, datapath: str = None, logger: JsonLogger = None, epochs: int = 1, lr: float = 1e-3,
                 wdk: float = 0.0, milestones: List[int] = (), batch_size: int = 128, ad_mode: str = "one_vs_rest",
                 device: Union[str, torch.device] = "cuda", oe_limit_samples=np.inf, oe_limit_classes=np.inf,
                 msms=(), workers: int = 2, classes: List[str] = None, data_parallel: bool = False,
                 graph_steps: bool = False, sync_bn: bool = True, exact_bn="auto"):
        """same parameters as the reference (`ad_trainer.py:98-164`).  `dataset` is either a step-batch source
        (eoe_amd.data: an object with `.loaders(batch_size)`, `.nominal_label`, `.normalize`) or a callable
        `(cls, seed) -> source`; `classes` names the classes to iterate (default: one class "0")."""
        self.model = model.cpu() if model is not None else model
        # replay forward + loss + backward + scores of the full-size step batch from a HIP graph (eoe_amd.GraphedStep): for the
        # launch-bound small encoders (CNN32 at 32x32); single GPU only, ragged batches run eagerly
        self.graph_steps = graph_steps
        self.train_transform, self.test_transform = train_transform, test_transform
        self.dsstr, self.oe_dsstr, self.datapath = dataset, oe_dataset, datapath
        self.logger = logger if logger is not None else JsonLogger(None)
        self.device = torch.device(device)
        self.epochs, self.lr, self.wdk, self.milestones, self.batch_size = epochs, lr, wdk, list(milestones), batch_size
        self.ad_mode = ad_mode
        self.center = None
        self.workers = workers
        self.ds = dataset if hasattr(dataset, "loaders") else None
        if classes is None:
            # a labelled multi-class set (eoe_amd.data.LabelledImageSet) names its own classes, as `str_labels(dsstr)` does
            # in the reference (ad_trainer.py:226); a single pre-built task is one class "0"
            classes = list(dataset.classes) if hasattr(dataset, "source") and hasattr(dataset, "classes") else ["0"]
        self.classes = classes
        self.data_parallel = data_parallel
        self.sync_bn = sync_bn      # data parallel only: BatchNorm statistics of the GLOBAL step batch (eoe_amd.parallel.enable_sync_bn)
        # BatchNorm encoders (CNN32 / CNN28 / WideResNet): the convolutions and linear layers as exact-fp32 implicit GEMMs on the fp32
        # matrix cores (eoe_amd.set_parity_mode, csrc/parity.hip).  The reference runs these nets in fp32 (models/cnn.py:73-86,
        # models/resnet.py:85-149); with 16-bit MFMA operands Adam at lr >= 1e-3 amplifies the operand rounding to 3-10 x the
        # reference's own fp32-vs-fp64 noise within ten steps (tests/test_gpu_parity_big.py), the fp32 mode stays inside it.
        # "auto" = on for a BatchNorm encoder trained with lr >= 1e-3 (every BatchNorm runner of the reference: train_cifar.py:17,
        # train_imagenet.py:16), True / False = forced.  The ViT has no BatchNorm and meets the bar in its 16-bit mode.
        self.exact_bn = exact_bn
        if msms:
            raise NotImplementedError("multi-scale modes (MSM) are out of scope")

    # ------------------------------------------------------------------------------------------- run
    def get_nominal_classes(self, cur_class: int):
        """the normal classes of the task "class `cur_class`" under the AD mode (ad_trainer.py:166-175)"""
        n = len(self.classes)
        if self.ad_mode == "one_vs_rest":
            return [cur_class]
        elif self.ad_mode == "leave_one_out":
            return [c for c in range(n) if c != cur_class]
        elif self.ad_mode == "fifty_fifty":
            return [c % n for c in range(cur_class, n // 2 + cur_class)]
        raise NotImplementedError(f"AD mode {self.ad_mode} unknown. Known modes are {ADTrainer.AD_MODES}.")

    def _dataset(self, c: int, seed: int):
        """the task of one (class, seed) run.  The reference builds it with `load_dataset(dsstr, datapath,
        self.get_nominal_classes(c), 0, ...)` unless `trainer.ds` was pre-set (ad_trainer.py:248-253): here a pre-built
        step-batch source is used as is, a labelled image set (`.source(normal_classes, seed)`) is asked for the task of the
        current AD mode, and a callable (cls, seed) -> source decides for itself."""
        if self.ds is not None:
            return self.ds
        if hasattr(self.dsstr, "source"):
            return self.dsstr.source(self.get_nominal_classes(c), seed)
        if callable(self.dsstr):
            return self.dsstr(c, seed)
        raise ValueError("dataset must be a step-batch source, a labelled image set or a callable (cls, seed) -> source")

    NAN_ATTEMPTS, NAN_GIVE_UP_AT = 5, 3      # ad_trainer.py:257-280: up to five tries; the third failure clears the result

    def _fresh_model(self, preset) -> torch.nn.Module:
        """the model one (class, seed) run starts from (ad_trainer.py:232-243): a module handed in through `load` as is,
        otherwise a copy of the CPU master with every layer re-initialised through its own `reset_parameters`"""
        if isinstance(preset, torch.nn.Module):
            model = deepcopy(preset)
        else:
            model = deepcopy(self.model)
            model.apply(weight_reset)
        for p in model.parameters():
            p.detach_().requires_grad_()
        return model

    def _train_with_retries(self, c: int, cstr: str, seed: int, preset, train: bool):
        """NaN scores abort a run (`NanGradientsError`); the reference then starts over on a freshly built dataset with fresh
        weights.  Returns (model, training ROC, dataset); the model is None only if the reference would return None"""
        ds = self._dataset(c, seed)
        model = roc = None
        for attempt in range(self.NAN_ATTEMPTS):
            model = self._fresh_model(preset)
            try:
                if train:
                    model, roc = self.train_cls(model, ds, c, cstr, seed, preset)
                return model, roc, ds
            except NanGradientsError:
                self.logger.warning(f'NaN scores while training class {c} "{cstr}", seed {seed} (failure {attempt + 1} of '
                                    f'{self.NAN_ATTEMPTS}); retrying with fresh weights and data.')
                ds = self._dataset(c, seed)
                if attempt + 1 == self.NAN_GIVE_UP_AT:
                    model = roc = None
        return model, roc, ds

    def run(self, run_classes: List[int] = None, run_seeds: int = 1, load: List[List] = None, test: bool = True,
            train: bool = True) -> Tuple[List[List[torch.nn.Module]], dict]:
        """every requested class x `run_seeds` repetitions (`ad_trainer.py:177-354`): train, evaluate, snapshot.
        Returns (models[class][seed], {'mean_auc', 'mean_avg_prec', 'std_auc', 'cls_aucs'})"""
        n_cls = len(self.classes)
        wanted = set(range(n_cls)) if run_classes is None else set(run_classes)
        # ad_trainer.py:231-232: a pre-loaded dataset is ONE task; iterating classes over it would train the same task again
        assert self.ds is None or len(wanted) == 1, "pre-loading DS (setting trainer.ds to something) only allowed for one class"
        models, train_rocs = [[] for _ in range(n_cls)], [[] for _ in range(n_cls)]
        eval_rocs, eval_prcs = [[] for _ in range(n_cls)], [[] for _ in range(n_cls)]
        for c, cstr in enumerate(self.classes):
            if c not in wanted:
                continue
            for seed in range(run_seeds):
                self.logger.print(f'------ start training cls {c} "{cstr}" ------')
                preset = None
                if load is not None and c < len(load) and seed < len(load[c]):
                    preset = load[c][seed]
                model, roc, ds = self._train_with_retries(c, cstr, seed, preset, train)
                train_rocs[c].append(roc)
                roc = prc = None
                if test and model is not None:
                    roc, prc = self.eval_cls(model, ds, c, cstr, seed)
                eval_rocs[c].append(roc)
                eval_prcs[c].append(prc)
                if model is not None:
                    self.logger.snapshot(f"snapshot_cls{c}_it{seed}", model, epoch=self.epochs,
                                         ds_statistics=getattr(ds, "ds_statistics", None))
                models[c].append(model if (model is None or ADTrainer.KEEP_SNAPSHOT_IN_RAM) else None)

        def per_class(curves, attr):
            """mean over the seeds of each class that produced a curve"""
            return [float(np.mean([getattr(r, attr) for r in lst if r is not None]))
                    for lst in curves if any(r is not None for r in lst)]

        mean_auc = std_auc = mean_avg_prec = float("nan")
        if test:
            aucs, aps = per_class(eval_rocs, "auc"), per_class(eval_prcs, "avg_prec")
            if aucs:
                mean_auc, std_auc = float(np.mean(aucs)), float(np.std(aucs))
            if aps:
                mean_avg_prec = float(np.mean(aps))
            self.logger.logtxt(f"Eval: Overall {mean_auc * 100:04.2f}% +- {std_auc * 100:04.2f}% AUC.")
        cls_aucs = [[None if r is None else r.get_score() for r in lst] for lst in eval_rocs]
        self.logger.logjson("results", {"eval_mean_auc": mean_auc, "eval_std_auc": std_auc,
                                        "eval_mean_avg_prec": mean_avg_prec, "eval_cls_rocs": cls_aucs,
                                        "classes": self.classes})
        return models, {"mean_auc": mean_auc, "mean_avg_prec": mean_avg_prec, "std_auc": std_auc, "cls_aucs": cls_aucs}

    # ------------------------------------------------------------------------------------------- hot loop
    def _normalize_hook(self, model, ds):
        """install the (mean, std) of the normal class on an encoder that fuses it; returns a fallback callable for
        encoders that do not"""
        norm = getattr(ds, "normalize", None)
        enc = getattr(model, "feature_model", model)
        if hasattr(enc, "set_normalize"):
            enc.set_normalize(*(norm if norm is not None else (None, None)))
            return None
        if norm is not None:
            raise NotImplementedError("this encoder has no fused normalise; add one to its first kernel")
        return None

    # the fp16 gradient scale (ops.set_grad_scale) follows the usual dynamic rule: halved when a step was dropped for non-finite
    # gradients, doubled again after SCALE_GROWTH_INTERVAL clean steps (never above SCALE_MAX or below 1).  The optimiser's device
    # counter is read every SCALE_POLL_EVERY steps and at the end of an epoch -- between an overflow and the next poll the steps
    # keep being dropped on the device, nothing non-finite reaches the weights or the moments.
    SCALE_POLL_EVERY, SCALE_GROWTH_INTERVAL, SCALE_MAX = 16, 2000, 65536.0

    def _move_grad_scale(self, opt, clean_steps: int, gstep: int, graphed):
        skipped = opt.skipped_steps() if hasattr(opt, "skipped_steps") else 0
        scale = ops.grad_scale()
        if skipped > 0:
            new = max(1.0, scale / 2.0)
            clean_steps = 0
        elif clean_steps >= self.SCALE_GROWTH_INTERVAL and 1.0 < scale < self.SCALE_MAX:
            new, clean_steps = scale * 2.0, 0
        else:
            return clean_steps, graphed
        if new != scale:
            ops.set_grad_scale(new)
            self.scale_events.append((gstep, new))
            self.logger.print(f"fp16 gradient scale {scale:g} -> {new:g} at step {gstep}" + (f" ({skipped} step(s) dropped)" if skipped else ""))
            graphed = None                   # a captured step has the old scale baked into its loss kernel: capture again
        return clean_steps, graphed

    def _exact_bn_for(self, model: torch.nn.Module) -> bool:
        if self.exact_bn is True or self.exact_bn is False:
            return bool(self.exact_bn)
        has_bn = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules())
        return has_bn and self.lr >= 1e-3

    def make_optimizer(self, model: torch.nn.Module):
        """Adam for every encoder (ad_trainer.py:383); the CLIP objective overrides this with SGD-Nesterov (:380-381)"""
        return FusedAdam(model.parameters(), lr=self.lr, weight_decay=self.wdk, amsgrad=False)

    def train_cls(self, model: torch.nn.Module, ds, cls: int, clsstr: str, seed: int,
                  load: Union[torch.nn.Module, str] = None):
        """the inner loop, `ad_trainer.py:356-471`; returns (model on the CPU in eval mode, training ROC)"""
        model = model.to(self.device).train()
        epochs = self.epochs
        cls_roc = None
        opt = self.make_optimizer(model)                                                              # :380-383
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, self.milestones, 0.1)                       # :384
        loader, _ = ds.loaders(self.batch_size, num_workers=self.workers, persistent=True)              # :385
        ep = self.load(load if isinstance(load, str) else None, model, opt, sched)                      # :396
        center = self.center = self.prepare_metric(clsstr, loader, model, seed)                         # :397
        self._normalize_hook(model, ds)
        rank, world = 0, 1
        arena = comm = None
        nominal = getattr(ds, "nominal_label", 0)
        self.last_losses = []
        self.last_scores = []                               # per epoch: (labels, scores) of every step batch, in order, on the device
        self.scale_events = []                              # (global step, new scale) whenever the fp16 gradient scale moved
        graphed = None                                      # (batch shape, GraphedStep) of the full-size step batch
        # process-wide state this loop changes (gradient scale, wgrad launch form, BatchNorm hook) is restored on every way out
        prev_scale = ops.grad_scale()
        prev_parity = ops.parity_mode()
        try:
            ops.set_parity_mode(self._exact_bn_for(model) or prev_parity)
            # fp16 compute: scale the loss gradient so that the 16-bit backward chain does not underflow (ops.set_grad_scale); the
            # optimiser un-scales, drops a step whose gradients overflowed (optim._NonFiniteGuard) and this loop moves the scale
            ops.set_grad_scale(ops.default_grad_scale())
            clean_steps, gstep = 0, 0
            if self.data_parallel and torch.distributed.is_initialized():
                rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
                # on RCCL: the library's own communicator (side HIP stream, reduce-scatter + all-gather, BatchNorm sums inside the
                # library); gloo rehearsals: torch.distributed collectives (parallel.make_comm)
                comm, _ = parallel.make_comm()
                arena = parallel.GradArena(model, comm=comm)
                arena.install_hooks()
                if any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules()):
                    # the reference's BatchNorm layers see the whole step batch: keep that meaning across the ranks
                    if self.sync_bn:
                        parallel.enable_sync_bn(comm=comm)
                    elif rank == 0:
                        self.logger.warning("data parallel training of a BatchNorm encoder with sync_bn=False: batch statistics are per "
                                            "rank, not those of the single-device full batch (eoe_amd/parallel.py)")
            for ep in range(ep, epochs):
                ep_labels, ep_scores, ep_losses = [], [], []
                for batch in loader:                                                                    # :410
                    imgs, lbls = batch[0], batch[1]
                    n_norm = int((lbls == nominal).sum())
                    n_glob = lbls.shape[0]
                    inv_count, keep_scores = 1.0 / n_glob, True
                    if world > 1:
                        halves = [h for h in (n_norm, n_glob - n_norm) if h > 0]
                        if min(halves) >= world:
                            rows = parallel.shard_rows(n_norm, n_glob - n_norm, rank, world)
                            imgs, lbls = imgs[rows], lbls[rows]
                        else:
                            # a ragged last batch with fewer rows than ranks in one half (no drop_last, bases.py:231-235): every
                            # rank computes the whole batch, weighted 1 / world, so that the summed gradient is the full-batch
                            # one and all ranks issue the same collectives; rank 0 alone reports its scores
                            inv_count, keep_scores = 1.0 / (n_glob * world), rank == 0
                    imgs = imgs.to(self.device, non_blocking=True)                                      # :411
                    lbls = lbls.to(self.device, non_blocking=True)                                      # :412
                    opt.zero_grad()                                                                     # :428
                    if self.graph_steps and world == 1 and graphed is None:
                        from ..graph import GraphedStep
                        graphed = (tuple(imgs.shape), GraphedStep(
                            model, lambda f, y: self.loss(f, y, center, inputs=None, nominal_label=nominal, inv_count=inv_count),
                            lambda f: self.compute_anomaly_score(f, center, inputs=None, nominal_label=nominal), imgs, lbls))
                    if graphed is not None and graphed[0] == tuple(imgs.shape):
                        loss, scores = graphed[1](imgs, lbls)                                           # :429-431,434 replayed
                        opt.step()                                                                      # :432
                        loss, scores = loss.detach().clone(), scores.detach().clone()                   # static graph outputs
                    else:
                        feats = model(imgs)                                                             # :429
                        loss = self.loss(feats, lbls, center, inputs=imgs, nominal_label=nominal,
                                         inv_count=inv_count)                                           # :430
                        loss.backward()                                                                 # :431
                        if arena is not None:
                            arena.finish()
                        opt.step()                                                                      # :432
                        opt.zero_grad()                                                                 # :433
                        scores = self.compute_anomaly_score(feats, center, inputs=imgs, nominal_label=nominal)   # :434
                    if keep_scores:
                        ep_labels.append(lbls)
                        ep_scores.append(scores.detach().reshape(-1))
                    ep_losses.append(loss.detach())
                    gstep += 1
                    clean_steps += 1
                    if gstep % self.SCALE_POLL_EVERY == 0:
                        clean_steps, graphed = self._move_grad_scale(opt, clean_steps, gstep, graphed)
                clean_steps, graphed = self._move_grad_scale(opt, clean_steps, gstep, graphed)
                # ---- epoch tail (:447-469): one host copy per epoch
                la = torch.cat(ep_labels) if ep_labels else torch.zeros(0, dtype=torch.int64, device=self.device)
                sc = torch.cat(ep_scores) if ep_scores else torch.zeros(0, dtype=torch.float32, device=self.device)
                ls = torch.stack(ep_losses)
                if world > 1:
                    la, sc = parallel.all_gather_1d(la), parallel.all_gather_1d(sc)
                    torch.distributed.all_reduce(ls)          # local losses are already divided by the global batch
                self.last_losses.extend(ls.cpu().tolist())
                self.last_scores.append((la, sc))
                if bool(torch.isnan(sc).any()):
                    raise NanGradientsError()                                                           # :448-449
                if bool((la == 1).any()):
                    # the epoch's AUC from the GPU-resident scores (eoe_auc_ap: exact pair counts), no host copy of the scores
                    cls_roc = ROC(auc_ap_device(la, sc)[0] if sc.is_cuda else roc_auc(la.cpu().numpy(), sc.cpu().numpy()))   # :452-455
                sched.step()                                                                            # :468
        finally:
            ops.set_grad_scale(prev_scale)
            ops.set_parity_mode(prev_parity)
            if arena is not None:
                parallel.disable_sync_bn()
                arena.remove_hooks()
                for p in model.parameters():
                    if hasattr(p, "_eoe_grad_buf"):
                        del p._eoe_grad_buf
            if comm is not None:
                comm.close()
        return model.cpu().eval(), cls_roc                                                              # :471

    def eval_cls(self, model: torch.nn.Module, ds, cls: int, clsstr: str, seed: int):
        """forward-only scoring of the test split, `ad_trainer.py:473-550`"""
        model = model.to(self.device).eval()
        _, loader = ds.loaders(self.batch_size, num_workers=self.workers, shuffle_test=False)
        self._normalize_hook(model, ds)
        center = self.center
        nominal = getattr(ds, "nominal_label", 0)
        ep_labels, ep_scores, ep_idcs = [], [], []
        prev_parity = ops.parity_mode()
        ops.set_parity_mode(self._exact_bn_for(model) or prev_parity)      # scored in the arithmetic it was trained in
        try:
            for batch in loader:
                imgs, lbls = batch[0].to(self.device), batch[1]
                with torch.no_grad():
                    feats = model(imgs)
                ep_scores.append(self.compute_anomaly_score(feats, center, inputs=imgs, nominal_label=nominal))
                ep_labels.append(lbls)
                ep_idcs.append(batch[2] if len(batch) > 2 else torch.arange(len(lbls)))
        finally:
            ops.set_parity_mode(prev_parity)
        la_t, sc_t = torch.cat(ep_labels), torch.cat(ep_scores).reshape(-1)
        la = la_t.cpu().numpy()
        sc = sc_t.cpu().numpy()                        # host copy only for the per-sample score log below
        idc = torch.cat(ep_idcs).cpu().numpy()
        if (la == 0).sum() > 0 and (la == 1).sum() > 0:
            if sc_t.is_cuda:                           # AUC / AP on the device (ad_trainer.py:517-521)
                auc, ap = auc_ap_device(la_t, sc_t)
            else:
                auc, ap = roc_auc(la, sc), average_precision(la, sc)
            cls_roc, cls_prc = ROC(auc), PRC(ap)
            self.logger.logtxt(f'Eval: class "{clsstr}" yields {cls_roc.auc * 100:04.2f}% AUC and '
                               f'{cls_prc.avg_prec * 100:04.2f}% average precision (seed {seed}).')
        else:
            cls_roc = cls_prc = None
        self.logger.logjson(f"eval_cls{cls}_it{seed}_anomaly_scores", {int(k): float(v) for k, v in zip(idc, sc)})
        model.cpu()
        return cls_roc, cls_prc

    # ------------------------------------------------------------------------------------------- snapshots
    def load(self, path: str, model: torch.nn.Module, opt=None, sched=None) -> int:
        """restore a snapshot file into (model, opt, sched) and return the epoch to resume from (`ad_trainer.py:552-598`).
        Two layouts are understood (`unify_snapshot_style`): the trainer's own {net, opt, sched, epoch, ...} and a bare
        state_dict, which is taken as pre-trained encoder weights of a `CustomNet`.  With or without a file, a model that can
        freeze its encoder does so here -- this is where the reference applies `freeze_parts` (:593-596)."""
        epoch = 0
        if path is not None:
            snap = self.unify_snapshot_style(torch.load(path, map_location="cpu"))
            encoder_weights = snap.get("feature_model")
            if encoder_weights is not None:
                if not hasattr(model, "load_feature_model_weights"):
                    raise ValueError(f"{path} holds encoder weights for a CustomNet, but the model to train is a "
                                     f"{type(model).__name__}, which has no feature model to load them into.")
                model.load_feature_model_weights(encoder_weights)
            for key, target in (("net", model), ("opt", opt), ("sched", sched)):
                state = snap.get(key)
                if state is not None and target is not None:
                    target.load_state_dict(state)
            epoch = snap.get("epoch") or 0
        if hasattr(model, "freeze_parts"):
            model.freeze_parts()
        return epoch

    def unify_snapshot_style(self, snapshot: dict) -> dict:
        """map both accepted file layouts onto one dict (`ad_trainer.py:608-615`)"""
        if isinstance(snapshot.get("net"), dict):
            return snapshot
        if snapshot and all(torch.is_tensor(v) for v in snapshot.values()):
            return {"feature_model": snapshot}
        raise ValueError("unrecognised snapshot layout: neither a trainer snapshot with a 'net' state_dict nor a bare state_dict")

    # ------------------------------------------------------------------------------------------- objective hooks
    @abstractmethod
    def prepare_metric(self, cstr: str, loader, model: torch.nn.Module, seed: int, **kwargs) -> torch.Tensor:
        pass

    @abstractmethod
    def compute_anomaly_score(self, features: torch.Tensor, center: torch.Tensor, **kwargs) -> torch.Tensor:
        pass

    @abstractmethod
    def loss(self, features: torch.Tensor, labels: torch.Tensor, center: torch.Tensor, **kwargs) -> torch.Tensor:
        pass
