"""ppsci.solver.Solver (/root/reference/ppsci/solver/solver.py:128-1116, train.py:58-213, eval.py,
printer.py) re-implemented over the fused HIP engine.

Constructor signature and the public methods (train / eval / predict / export stubs) follow the reference.
What differs by design: every constraint is compiled ONCE into (Taylor-mode stream set, epilogue
program) and a training iteration is a fixed sequence of kernel launches (engine.Engine); loss values
are only read back (one device->host sync) every `log_freq` iterations, whereas the reference syncs on
`.item()` for every loss term of every iteration (expression.py:122, train.py:145).
Visualizers (`visualizer=`, `visualize()`) evaluate their expressions through `predict` and hand the arrays to
`ppsci.visualize`'s writers.  Not supported (raise): AMP, to_static, loss aggregators other than Sum / GradNorm / NTK."""
from __future__ import annotations

import datetime
import os
import time
from typing import Any, Callable, Dict, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist

from .. import autodiff
from .. import hotpath as hp
from ..compile import CompiledConstraint
from ..device import get_device
from ..engine import Engine
from ..loss import mtl
from ..utils import logger, misc, save_load


def _metric_value(v):
    """A scalar metric as a float; `keep_batch=True` metrics (one value per sample) are reported by their mean."""
    t = torch.as_tensor(v).detach().float()
    return float(t) if t.numel() == 1 else float(t.mean())


def _is_full_static_batch(cst) -> bool:
    ds = getattr(cst.data_loader, "dataset", cst.data_loader)
    if getattr(ds, "is_iterable", False):
        return type(ds).__name__ == "IterableNamedArrayDataset"
    return len(cst.data_loader) == 1 and not cst.data_loader.batch_sampler.shuffle


class Solver:
    def __init__(
        self,
        model,
        constraint: Optional[Dict[str, Any]] = None,
        output_dir: Optional[str] = "./output/",
        optimizer=None,
        lr_scheduler=None,
        epochs: int = 5,
        iters_per_epoch: int = 20,
        update_freq: int = 1,
        save_freq: int = 0,
        log_freq: int = 10,
        eval_during_train: bool = False,
        start_eval_epoch: int = 1,
        eval_freq: int = 1,
        seed: int = 42,
        use_vdl: bool = False,
        use_wandb: bool = False,
        use_tbd: bool = False,
        wandb_config: Optional[Mapping] = None,
        device: str = "gpu",
        equation: Optional[Dict[str, Any]] = None,
        geom: Optional[Dict[str, Any]] = None,
        validator: Optional[Dict[str, Any]] = None,
        visualizer: Optional[Dict[str, Any]] = None,
        use_amp: bool = False,
        amp_level: str = "O1",
        pretrained_model_path: Optional[str] = None,
        checkpoint_path: Optional[str] = None,
        compute_metric_by_batch: bool = False,
        eval_with_no_grad: bool = False,
        to_static: bool = False,
        loss_aggregator: Optional[mtl.LossAggregator] = None,
        *,
        cfg=None,
        dp_reduce: str = "sum",
    ):
        if use_amp or to_static:
            raise NotImplementedError("AMP / to_static are not available on the fused HIP path (fp32 only)")
        if int(update_freq) < 1:
            raise ValueError(f"update_freq should be a positive integer, but got {update_freq}")
        if loss_aggregator is not None and not (isinstance(loss_aggregator, mtl.Sum)
                                                or getattr(loss_aggregator, "per_loss_grad", False)):
            raise NotImplementedError("loss aggregators on the fused path: Sum, GradNorm, NTK")
        self.cfg = cfg
        if cfg is not None and hasattr(cfg, "get") and cfg.get("TRAIN", None) is not None or (
                cfg is not None and hasattr(cfg, "get") and cfg.get("output_dir", None) is not None):
            # `Solver(model, constraint, ..., cfg=cfg)`: the run parameters come from the config, as in the reference
            # (/root/reference/ppsci/solver/solver.py:165-168, _parse_params_from_cfg :1078-1116), with the defaults its config schema
            # fills in (ppsci/utils/config.py TrainConfig / EvalConfig / SolverConfig)
            def sub(name):
                v = cfg.get(name, None)
                return v if v is not None else {}

            tr, ev = sub("TRAIN"), sub("EVAL")
            if cfg.get("use_amp", False) or cfg.get("to_static", False):
                raise NotImplementedError("AMP / to_static are not available on the fused HIP path (fp32 only)")
            output_dir = cfg.get("output_dir", output_dir)
            log_freq = cfg.get("log_freq", 20)
            seed = cfg.get("seed", 42)
            epochs = tr.get("epochs", epochs)
            iters_per_epoch = tr.get("iters_per_epoch", 20)
            update_freq = tr.get("update_freq", 1)
            save_freq = tr.get("save_freq", 0)
            eval_during_train = tr.get("eval_during_train", False)
            start_eval_epoch = tr.get("start_eval_epoch", 1)
            eval_freq = tr.get("eval_freq", 1)
            checkpoint_path = tr.get("checkpoint_path", None)
            compute_metric_by_batch = ev.get("compute_metric_by_batch", False)
            eval_with_no_grad = ev.get("eval_with_no_grad", False)
            mode = cfg.get("mode", "train")
            pretrained_model_path = (tr if mode == "train" else ev if mode == "eval" else sub("INFER")).get("pretrained_model_path", None)
        self.model = model
        self.constraint = constraint
        self.output_dir = output_dir
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.epochs = epochs
        self.iters_per_epoch = iters_per_epoch
        self.update_freq = update_freq
        self.save_freq = save_freq
        self.log_freq = log_freq
        self.eval_during_train = eval_during_train
        self.start_eval_epoch = start_eval_epoch
        self.eval_freq = eval_freq
        self.seed = seed
        self.equation = equation
        self.geom = geom
        self.validator = validator
        self.visualizer = visualizer
        self.compute_metric_by_batch = compute_metric_by_batch
        self.eval_with_no_grad = eval_with_no_grad
        self.loss_aggregator = loss_aggregator or mtl.Sum()
        self.vdl_writer = self.wandb_writer = self.tbd_writer = None
        self.benchmark_flag = bool(os.getenv("BENCHMARK_ROOT", None))
        self.device = get_device()
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.global_step = 0
        self.best_metric = {"metric": float("inf"), "epoch": 0}
        self.train_output_info: Dict[str, misc.AverageMeter] = {}
        self.train_loss_info: Dict[str, list] = {}
        self.train_time_info = {"batch_cost": misc.AverageMeter("batch_cost", ".5f", postfix="s"),
                                "reader_cost": misc.AverageMeter("reader_cost", ".5f", postfix="s")}
        self.eval_output_info: Dict[str, misc.AverageMeter] = {}
        self.eval_time_info = {"batch_cost": misc.AverageMeter("batch_cost", ".5f", postfix="s"),
                               "reader_cost": misc.AverageMeter("reader_cost", ".5f", postfix="s")}

        if pretrained_model_path is not None:
            save_load.load_pretrain(self.model, pretrained_model_path, self.equation)
        if checkpoint_path is not None:
            self.best_metric = save_load.load_checkpoint(checkpoint_path, self.model, self.optimizer, self.equation,
                                                           aggregator=self.loss_aggregator)

        if self.world_size > 1:
            # DataParallel wrap of the reference (solver.py:388-412): replicate rank 0's parameters
            dist.broadcast(self.model.flat_params, src=0)
        from ..arch.spinn import SPINN

        self._is_spinn = isinstance(self.model, SPINN)
        self._is_operator = bool(getattr(self.model, "is_operator", False))
        if self._is_spinn:
            from ..spinn_engine import SpinnEngine

            self.engine = SpinnEngine(self.model)
        elif self._is_operator:
            from ..operator_engine import OperatorEngine

            self.engine = OperatorEngine(self.model)
        else:
            self.engine = Engine(getattr(self.model, "layout", None), self.model.kernel_params, dp_reduce=dp_reduce)
        # factored / tied layers (weight_norm, random_weight, fourier): the kernels read model.kernel_params,
        # rebuilt from the trainable tensors before every sweep; their gradient is pulled back afterwards
        self._reparam = bool(getattr(self.model, "reparam", False))
        self._acc_grad, self._acc_count = None, 0  # gradient accumulation (update_freq > 1)
        self.latest_save_interval = float(os.environ.get("PPSCI_LATEST_SAVE_INTERVAL", "1.0"))  # seconds; 0 = every epoch
        self._latest_saved_at = float("-inf")
        if (self.optimizer is not None and not self._is_spinn and not self._is_operator and not self._reparam
                and hasattr(self.optimizer, "beta1")):
            self.engine.m, self.engine.v = self.optimizer.m, self.optimizer.v
            self.engine.beta1, self.engine.beta2, self.engine.eps = (self.optimizer.beta1, self.optimizer.beta2,
                                                                    self.optimizer.epsilon)
            self.engine.t = self.optimizer.t

        # ---- compile constraints (convert_expr, solver.py:496-535)
        self._compiled: Dict[str, CompiledConstraint] = {}
        self._static: Dict[str, bool] = {}
        self._ragged: set = set()  # constraints whose sample count does not divide over the ranks (zero-weight padding)
        self._device_data: Dict[str, Tuple[dict, dict, dict]] = {}
        if self.constraint:
            for name, cst in self.constraint.items():
                self._compiled[name] = self._compile_constraint(name, cst)
        self._compiled_val: Dict[str, CompiledConstraint] = {}
        self._predict_cache: Dict[tuple, CompiledConstraint] = {}

    # ------------------------------------------------------------------ compilation helpers
    def _equation_exprs(self, exprs: Dict[str, Callable]) -> Dict[str, Callable]:
        return dict(exprs)

    def _compile_spinn_constraint(self, name: str, cst):
        """Separable nets: the expression must be linear in {u, u_xx, u_yy, u_zz} (arch.spinn.GridLinear)."""
        from ..arch.spinn import GridLinear
        from ..graph import Sym
        from ..spinn_engine import SpinnConstraint

        ds = getattr(cst.data_loader, "dataset", cst.data_loader)
        if len(ds.label_keys) != 1:
            raise NotImplementedError("one label key per SPINN constraint")
        key = ds.label_keys[0]
        data = {k: Sym.input(k) for k in self.model.input_keys}
        data.update(self.model(data))
        val = cst.output_expr[key](data) if key in cst.output_expr else data[key]
        if not isinstance(val, GridLinear):
            raise NotImplementedError(f"constraint {name}: expression {key!r} is not a linear form of the SPINN output")
        loss = cst.loss
        sc = SpinnConstraint(name, self.model, val.c, key, lambda total, k=key: loss.term_scale(k, total), self.device,
                             self.world_size, self.rank)
        sc.batch_size = 0
        sc.label_keys = [key]
        return sc

    def _compile_constraint(self, name: str, cst) -> CompiledConstraint:
        if self._is_spinn:
            self._static[name] = False
            return self._compile_spinn_constraint(name, cst)
        if self._is_operator:
            from ..operator_engine import OperatorConstraint

            ds = getattr(cst.data_loader, "dataset", cst.data_loader)
            self._static[name] = False
            bsz = getattr(getattr(cst.data_loader, "batch_sampler", None), "batch_size", 0) or 0
            return OperatorConstraint(name, self.model, cst.output_expr, cst.loss, self.device, list(ds.label_keys), bsz)
        ds = getattr(cst.data_loader, "dataset", cst.data_loader)
        if getattr(ds, "shard_in_engine", False) and self.world_size > 1:
            raise NotImplementedError(f"constraint {name}: 'shard_in_engine' datasets are only sharded by the SPINN engine")
        input_keys = list(ds.input_keys)
        label_keys = list(ds.label_keys)
        if hasattr(ds, "weight_fn"):  # ContinuousNamedArrayDataset
            w0 = ds.weight_fn(ds.input_fn()) if callable(ds.weight_fn) else None
            weight_keys = list(w0.keys()) if w0 else []
        else:
            weight_keys = list((ds.weight or {}).keys())
        if getattr(ds, "is_iterable", False):
            inp, lab, w = next(iter(ds))
            bsz = len(next(iter(inp.values())))
        else:
            bsz = cst.data_loader.batch_sampler.batch_size
            per_rank = cst.data_loader.batch_sampler.num_samples  # ceil(n / world): what THIS rank's sampler yields
            if cst.data_loader.batch_sampler.drop_last is False and per_rank % bsz != 0 and len(cst.data_loader) > 1:
                raise NotImplementedError(f"constraint {name}: ragged last batch ({per_rank} % {bsz} != 0 samples per "
                                          "rank); use drop_last")
            if len(cst.data_loader) == 1:
                bsz = cst.data_loader.batch_sampler.num_samples
            if self.world_size > 1 and cst.data_loader.batch_sampler.n % self.world_size != 0:
                # Ragged shards: the sampler pads by wrapping around (data.BatchSampler, as the reference's
                # DistributedBatchSampler does) so that all ranks run the same batches; the duplicates get ZERO weight and
                # "mean" is taken over the true sample count, so the W-rank step equals the 1-rank step.
                # That needs a weight column for every loss key (_shard_weights fills it at bind time).
                if (getattr(cst.loss, "periodic", False) or getattr(cst.loss, "causal", None)
                        or hasattr(cst.loss, "batch_weight") or getattr(cst.loss, "term_kind", 0) != 0):
                    raise NotImplementedError(f"constraint {name}: {len(ds)} samples do not divide over {self.world_size} ranks; "
                                              f"zero-weight padding is built for MSELoss only, not {type(cst.loss).__name__}")
                self._ragged.add(name)
                weight_keys = weight_keys + [k for k in label_keys if k not in weight_keys]
        # A static constraint (one full batch, not shuffled) is bound ONCE, here: its expressions may then look at the values
        # of that batch (Python control flow on the input columns: graph.batch_values)
        self._static[name] = _is_full_static_batch(cst)
        first = next(cst.data_iter) if self._static[name] else None
        try:
            cc = CompiledConstraint(name, self.model, cst.output_expr, input_keys, label_keys, weight_keys, cst.loss, bsz,
                                    bsz * self.world_size, self.device, train=True,
                                    extra_parameters=self._extra_parameters(),
                                    static_batch=first[0] if first is not None else None)
        except (NotImplementedError, TypeError) as e:
            # the expressions are not a per-point program (arithmetic on row windows, tensor methods that reduce over the
            # batch, control flow on values that change every step, a derivative set beyond the instantiated stream sets
            # ...): refused with the reason -- there is no op-by-op fallback.  The other ranks are waiting in the collective
            # of _check_trace_decisions: take part in it (it raises there as well), then raise the reason of this rank
            self._check_trace_decisions(name, None, f"{type(e).__name__}: {e}")
            raise NotImplementedError(f"constraint {name}: not lowerable to the fused HIP kernels ({type(e).__name__}: {e})") from e
        if name in self._ragged and (cc.low.reductions or cc.low.couplings):
            raise NotImplementedError(f"constraint {name}: batch reductions / couplings over ragged data-parallel shards (the sampler's "
                                      "wrap-around duplicates would be counted twice); make the sample count a multiple of the ranks")
        if cc.specialised_to:
            logger.info(f"constraint {name}: traced for the values of its (fixed) batch: {', '.join(cc.specialised_to)}")
        self._check_trace_decisions(name, cc)
        if first is not None:
            inp, lab, w = first
            cc.bind(inp, lab, self._shard_weights(name, cst, lab, w))
        return cc

    def _shard_weights(self, name: str, cst, lab, w):
        """Zero weight for the wrap-around duplicates of a ragged shard (data.BatchSampler.last_pad), and, for "mean"
        losses, the other weights scaled so that the compiled 1 / (batch x world) becomes 1 / (true sample count)."""
        if name not in self._ragged:
            return w
        pad = getattr(cst.data_loader, "last_pad", None)
        out = dict(w or {})
        n_loc = len(next(iter(lab.values())))
        for k in lab:
            col = np.ones((n_loc, 1), np.float32) if k not in out else np.array(
                out[k].cpu().numpy() if isinstance(out[k], torch.Tensor) else out[k], dtype=np.float32).reshape(n_loc, 1)
            if pad is not None:
                mask, npad, nglob = pad
                col[mask] = 0.0
                if getattr(cst.loss, "reduction", "mean") == "mean":
                    col *= np.float32(nglob / (nglob - npad))
            out[k] = col
        return out

    # ------------------------------------------------------------------ training
    def train(self) -> None:
        """solver.py:544-669 + train.py:58-213."""
        if self.optimizer is None:
            raise ValueError("Solver.train needs an optimizer")
        self.global_step = self.best_metric["epoch"] * self.iters_per_epoch
        start_epoch = self.best_metric["epoch"] + 1
        csts = list(self._compiled.values())
        if getattr(self.loss_aggregator, "per_loss_grad", False):
            self._apply_loss_weights()
        total_batch_size = sum(c.batch_size for c in csts)
        for epoch_id in range(start_epoch, self.epochs + 1):
            batch_tic = time.perf_counter()
            for iter_id in range(1, self.iters_per_epoch + 1):
                reader_tic = time.perf_counter()
                for name, cc in self._compiled.items():
                    if not self._static[name]:
                        inp, lab, w = next(self.constraint[name].data_iter)
                        if self._is_spinn:
                            cc.bind(inp, lab)
                        else:
                            cc.bind(inp, lab, self._shard_weights(name, self.constraint[name], lab, w))
                reader_cost = time.perf_counter() - reader_tic
                eng_csts = csts if (self._is_spinn or self._is_operator) else [c.fused for c in csts]
                gscale = (1.0 / self.world_size) if (self.engine.dp_reduce == "mean" and self.world_size > 1) else 1.0
                if getattr(self.optimizer, "is_lbfgs", False):
                    # train_LBFGS_epoch_func (solver/train.py:216-315): the optimizer re-evaluates loss + gradient
                    def closure():
                        self._materialize()
                        self.engine.forward_backward(eng_csts)
                        self.engine.allreduce()
                        self._update_train_loss()
                        total = self.last_losses["loss"]
                        if self.world_size > 1:  # the line search must see the same value on every rank
                            tt = torch.tensor([total], dtype=torch.float64, device=self.device)
                            dist.all_reduce(tt)
                            total = float(tt[0]) * (gscale if gscale != 1.0 else 1.0)
                        g = self._train_grad()
                        return total, g * gscale if gscale != 1.0 else g

                    self.optimizer.step(closure)
                elif self._step_in_one_launch(eng_csts, gscale):
                    pass  # forward -> loss -> backward -> Adam of every constraint in one launch each (engine.step_one_launch)
                elif self._reduce_and_adam_in_one_launch(eng_csts, gscale):
                    pass  # (SPINN / FNO) the reductions that end the backward pass and the Adam update in one launch
                else:
                    self._materialize()
                    self.engine.forward_backward(eng_csts)
                    self.engine.allreduce()
                    self._allreduce_eq_params()
                    if getattr(self.loss_aggregator, "per_loss_grad", False):
                        # GradNorm / NTK: the step above used the current weights; refresh them from the per-key
                        # gradient norms at the same parameters (the total gradient is recomputed afterwards)
                        if self.loss_aggregator.needs_update(self.global_step):
                            self._update_loss_weights(eng_csts)
                    if self.update_freq > 1:
                        # train.py:141-142, :163-180: every loss is divided by update_freq, gradients accumulate and
                        # the optimizer steps every update_freq-th iteration and at the end of the epoch
                        if getattr(self.optimizer, "eq_store", None) is not None:
                            raise NotImplementedError("update_freq > 1 together with learnable equation parameters")
                        g = self._train_grad()
                        if self._acc_grad is None:
                            self._acc_grad = torch.zeros_like(g)
                        hp.reduce_rows(g.view(1, -1), 1, g.numel(), self._acc_grad, self._acc_count > 0)
                        self._acc_count += 1
                        if iter_id % self.update_freq == 0 or iter_id == self.iters_per_epoch:
                            self.optimizer.step(self._acc_grad, gscale / self.update_freq)
                            self._acc_count = 0
                    elif not self._adam_behind_allreduce(eng_csts, gscale):
                        self.optimizer.step(self._train_grad(), gscale)
                self.optimizer.clear_grad()
                if self.lr_scheduler is not None and not getattr(self.lr_scheduler, "by_epoch", False):
                    self.lr_scheduler.step()
                self.global_step += 1
                # The reference converts every loss term to a Python float in every iteration (expression.py:122), i.e. its
                # batch_cost always contains the device time.  Here the loss terms are fetched only when they are logged; that
                # fetch is the iteration's device->host sync, so it has to sit INSIDE the timed window: the logging iteration
                # then absorbs the queued device work of its window and the window's mean batch_cost / ips are true device
                # figures (a fetch after the clock stop reported launch time only: ips 8e8 for a 3 ms step).
                log_now = iter_id == 1 or iter_id % self.log_freq == 0
                if log_now:
                    self._update_train_loss()
                elif self.benchmark_flag and torch.cuda.is_available():
                    torch.cuda.synchronize()
                batch_cost = time.perf_counter() - batch_tic
                self.train_time_info["reader_cost"].update(reader_cost)
                self.train_time_info["batch_cost"].update(batch_cost)
                if log_now:
                    self._log_train_info(total_batch_size, epoch_id, iter_id)
                batch_tic = time.perf_counter()
            if self.lr_scheduler is not None and getattr(self.lr_scheduler, "by_epoch", False):
                self.lr_scheduler.step()
            cur_metric = float("inf")
            if self.eval_during_train and epoch_id % self.eval_freq == 0 and epoch_id >= self.start_eval_epoch:
                cur_metric, metric_dict_group = self.eval(epoch_id)
                if cur_metric < self.best_metric["metric"]:
                    self.best_metric["metric"] = cur_metric
                    self.best_metric["epoch"] = epoch_id
                    save_load.save_checkpoint(self.model, self.optimizer, self.best_metric, None, self.output_dir,
                                              "best_model", self.equation, aggregator=self.loss_aggregator)
                logger.info(f"[Eval][Epoch {epoch_id}][best metric: {self.best_metric['metric']}]")
                if self.visualizer is not None:  # "visualize after evaluation" (solver.py:602-604)
                    self.visualize(epoch_id)
            if self.save_freq > 0 and epoch_id % self.save_freq == 0:
                save_load.save_checkpoint(self.model, self.optimizer, {"metric": cur_metric, "epoch": epoch_id}, None,
                                          self.output_dir, f"epoch_{epoch_id}", self.equation, aggregator=self.loss_aggregator)
            # "always save the latest model for convenient resume training" (solver.py of the reference, every epoch).
            # With iters_per_epoch = 1 (laplace2d.yaml) an epoch is 30 us of GPU work and the three files cost
            # 0.5 ms, so `latest` is refreshed at most every `latest_save_interval` seconds and at the last epoch.
            now = time.perf_counter()
            if epoch_id == self.epochs or now - self._latest_saved_at >= self.latest_save_interval:
                self._latest_saved_at = now
                save_load.save_checkpoint(self.model, self.optimizer, {"metric": cur_metric, "epoch": epoch_id}, None,
                                          self.output_dir, "latest", self.equation,
                                          print_log=(epoch_id == self.epochs), aggregator=self.loss_aggregator)

    # ------------------------------------------------------------------ per-loss gradient weighting (GradNorm / NTK)
    def _loss_key_order(self):
        keys = []
        for cc in self._compiled.values():
            for k in cc.label_keys:
                if k not in keys:
                    keys.append(k)
        return keys

    def _apply_loss_weights(self, mask_key=None):
        """Residual scale = base scale x weight of its key (or, for a masked pass, base scale for one key and 0)."""
        agg = self.loss_aggregator
        keys = self._loss_key_order()
        for cc in self._compiled.values():
            if not hasattr(cc, "_base_scales"):
                cc._base_scales = [cc.fused.edesc.res[i].scale for i in range(len(cc.label_keys))]
            for i, k in enumerate(cc.label_keys):
                if mask_key is None:
                    cc.fused.edesc.res[i].scale = cc._base_scales[i] * float(agg.weight[keys.index(k)])
                else:
                    cc.fused.edesc.res[i].scale = cc._base_scales[i] if k == mask_key else 0.0
        self.engine.invalidate_graphs()

    def _extra_parameters(self):
        """solver.py:490-494 of the reference: the learnable parameters of every equation go to lambdify."""
        out = []
        for eq in (self.equation or {}).values():
            out += list(getattr(eq, "learnable_parameters", []))
        return out

    def _allreduce_eq_params(self) -> None:
        if self.world_size > 1 and getattr(self.optimizer, "eq_store", None) is not None:
            dist.all_reduce(self.optimizer.eq_store.grad)

    def _materialize(self) -> None:
        if self._reparam:
            self.model.materialize()

    def _check_trace_decisions(self, name: str, cc, failure: Optional[str] = None) -> None:
        """Python control flow on the values of a fixed batch is followed at trace time (graph.batch_values).  Under data
        parallelism every rank traces on ITS shard: the ranks must end up with the SAME program (residual program, loss terms,
        derivative streams), otherwise they would train different programs against one all-reduced gradient without anybody
        noticing (the reference evaluates the user's function on each rank's tensors every step, utils/expression.py:96-102, so
        there a rank-dependent branch is at least visible in the loss).  A collective: every rank calls it for every
        constraint, in the same order."""
        import zlib

        dist = torch.distributed
        if self.world_size <= 1 or not dist.is_available() or not dist.is_initialized():
            return
        if cc is None:  # this rank's trace raised: say so to everybody instead of leaving them in the collective
            mine = (None, None, [], failure or "trace failed")
        else:
            mine = (zlib.crc32(bytes(cc.fused.edesc)), repr(cc.fused.streams), list(cc.specialised_to), None)
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        failed = [(r, e[3]) for r, e in enumerate(everyone) if e[3] is not None]
        if failed:
            if cc is None:
                return  # the caller raises this rank's own reason
            raise NotImplementedError(f"constraint {name}: not lowerable on rank {failed[0][0]} ({failed[0][1]}); its shard takes "
                                      "a path through the expressions that this rank's shard does not")
        if any(e[:2] != everyone[0][:2] for e in everyone):
            odd = next(r for r, e in enumerate(everyone) if e[:2] != everyone[0][:2])
            msg = (f"constraint {name}: the expressions branch on values of the batch and ranks 0 and {odd} took different "
                   f"branches (rank 0 asked {everyone[0][2]}; rank {odd} asked {everyone[odd][2]}): their programs differ")
            if os.environ.get("PPSCI_RANK_SPECIFIC_TRACES", "0") == "1":
                logger.warning(msg + " -- accepted (PPSCI_RANK_SPECIFIC_TRACES=1): every rank trains its own program")
            else:
                raise RuntimeError(msg + "; make the condition independent of the shard, or set PPSCI_RANK_SPECIFIC_TRACES=1 "
                                         "to train rank-specific programs")

    def _step_in_one_launch(self, eng_csts, gscale: float) -> bool:
        """The whole iteration (train.py:82-184) as one launch per constraint, the optimizer step inside the last one,
        when nothing sits between the gradient and the update: one rank, plain Adam (no clipping / decay / learnable
        equation parameters), no re-parametrised weights, no gradient accumulation or per-loss gradients, and every
        constraint small enough for the one-launch kernel (engine.one_launch_ready).  False: nothing was done."""
        opt = self.optimizer
        if (self.world_size != 1 or not eng_csts or self._reparam or self.update_freq > 1
                or getattr(self.loss_aggregator, "per_loss_grad", False) or type(opt).__name__ != "_AdamState"
                or opt.grad_clip is not None or opt.l2 != 0.0 or opt.eq_store is not None
                or not hasattr(self.engine, "one_launch_ready") or opt.model.flat_params.data_ptr() != self.engine.params.data_ptr()
                or not self.engine.one_launch_ready(eng_csts)):
            return False
        self._materialize()
        opt.t += 1
        self.engine.step_one_launch(eng_csts, dict(m=opt.m, v=opt.v, lr=opt.get_lr(), beta1=opt.beta1, beta2=opt.beta2,
                                                   eps=opt.epsilon, grad_scale=gscale, t=opt.t))
        return True

    def _adam_behind_allreduce(self, eng_csts, gscale: float) -> bool:
        """Data parallelism, fused tile kernel (one constraint of padded width 64, plain Adam): the optimizer step and the next
        step's weight fragments in ONE launch behind the all-reduce (engine.apply_adam_fused) instead of the optimizer's own
        kernel + a weight-split launch.  False: nothing was done."""
        opt = self.optimizer
        if (self.world_size == 1 or self._reparam or type(opt).__name__ != "_AdamState" or opt.grad_clip is not None or opt.l2 != 0.0
                or opt.eq_store is not None or not hasattr(self.engine, "apply_adam_fused")
                or opt.model.flat_params.data_ptr() != self.engine.params.data_ptr()):
            return False
        if not self.engine.apply_adam_fused(eng_csts, dict(m=opt.m, v=opt.v, lr=opt.get_lr(), beta1=opt.beta1, beta2=opt.beta2,
                                                           eps=opt.epsilon, grad_scale=gscale, t=opt.t + 1)):
            return False
        opt.t += 1
        return True

    def _reduce_and_adam_in_one_launch(self, eng_csts, gscale: float) -> bool:
        """Engines whose backward pass ends in row reductions (SPINN: gradient rows of the branch nets + loss rows; FNO: the
        partials of the 1x1 convolutions' weight gradients): one rank, plain Adam, nothing between the gradient and the update
        -- the sums and the update are ONE launch (hp.reduce_rows_multi_adam) instead of two.  False: nothing was done."""
        opt, eng = self.optimizer, self.engine
        if (self.world_size != 1 or not hasattr(eng, "forward_backward_deferred") or type(opt).__name__ != "_AdamState"
                or opt.grad_clip is not None or opt.l2 != 0.0 or opt.eq_store is not None or self.update_freq > 1 or self._reparam
                or getattr(self.loss_aggregator, "per_loss_grad", False) or os.environ.get("PPSCI_FUSED_REDUCE_ADAM", "1") == "0"
                or opt.model.flat_params.data_ptr() != self.model.flat_params.data_ptr()):
            return False
        from ..engine import step_with_adam

        self._materialize()
        step_with_adam(eng, eng_csts, opt, self.model.flat_params, gscale)
        return True

    def _train_grad(self) -> torch.Tensor:
        """Gradient w.r.t. the optimizer's (trainable) parameters."""
        return self.model.pull_back(self.engine.grad) if self._reparam else self.engine.grad

    def _update_loss_weights(self, eng_csts):
        if self._is_spinn or self._is_operator:
            raise NotImplementedError("GradNorm / NTK need the fused PINN engine")
        saved = self.engine.grad.clone()
        saved_terms = [cc.fused.loss_terms.clone() for cc in self._compiled.values()]
        norms = []
        for k in self._loss_key_order():
            self._apply_loss_weights(mask_key=k)
            self.engine.forward_backward(eng_csts)
            self.engine.allreduce()
            norms.append(float(torch.linalg.norm(self._train_grad())))
        self.loss_aggregator.update(norms)
        self._apply_loss_weights()
        # gradient of THIS step's weighted loss (old weights), which is what the optimizer consumes; the masked
        # passes overwrote the loss terms too, but those are re-evaluated by the next step before being logged
        self.engine.grad.copy_(saved)
        for cc, t in zip(self._compiled.values(), saved_terms):
            cc.fused.loss_terms.copy_(t)

    def _update_train_loss(self):
        """printer.update_train_loss: total `loss` = Sum aggregator over all terms (mtl/sum.py:45-60), plus one
        entry per constraint = sum of its keys (expression.py:120-126)."""
        losses_all: Dict[str, float] = {}
        per_cst: Dict[str, float] = {}
        for name, cc in self._compiled.items():
            if self._is_operator:
                vals = cc.losses()  # keys are whatever the loss returns (FunctionalLoss), not the label keys
                keys = list(vals.keys())
            else:
                vals = {cc.label_key: cc.loss()} if self._is_spinn else cc.fused.losses()
                keys = cc.label_keys
                if getattr(self.loss_aggregator, "per_loss_grad", False) and hasattr(cc, "_base_scales"):
                    # the kernels applied the aggregator's weights through the residual scales: report raw terms
                    order = self._loss_key_order()
                    vals = {k: vals[k] / max(float(self.loss_aggregator.weight[order.index(k)]), 1e-30) for k in keys}
            per_cst[name] = 0.0
            for k in keys:
                per_cst[name] += vals[k]
                losses_all[k] = losses_all.get(k, 0.0) + vals[k]
        total = float(self.loss_aggregator(losses_all, self.global_step))
        if self.update_freq > 1:
            total /= self.update_freq  # train.py:141-142: the logged total is the scaled one
        self.last_losses = {"loss": total, **per_cst}
        for k, v in self.last_losses.items():
            if k not in self.train_output_info:
                self.train_output_info[k] = misc.AverageMeter(k, "7.5f")
            self.train_output_info[k].update(v, 1)
            # plot_loss_history: the reference keeps EVERY iteration's value (it synchronises on .item() every iteration,
            # train.py:145); here a value exists where the loss was fetched (every `log_freq` iterations): (iteration, value)
            self.train_loss_info.setdefault(k, []).append((self.global_step, v))

    def _log_train_info(self, batch_size: int, epoch_id: int, iter_id: int):
        lr_msg = f"lr: {self.optimizer.get_lr():.5f}"
        metric_msg = ", ".join(f"{k}: {m.avg:.5f}" for k, m in self.train_output_info.items())
        time_msg = ", ".join(m.mean for m in self.train_time_info.values())
        avg = self.train_time_info["batch_cost"].avg
        ips_msg = f"ips: {batch_size / avg:.2f}" + (" samples/s" if self.benchmark_flag else "")
        eta = ((self.epochs - epoch_id + 1) * self.iters_per_epoch - iter_id) * avg
        ew, iw = len(str(self.epochs)), len(str(self.iters_per_epoch))
        logger.info(f"[Train][Epoch {epoch_id:>{ew}}/{self.epochs}][Iter {iter_id:>{iw}}/{self.iters_per_epoch}] {lr_msg}, "
                    f"{metric_msg}, {time_msg}, {ips_msg}, eta: {str(datetime.timedelta(seconds=int(eta)))}")
        for m in self.train_time_info.values():
            m.reset()
        for m in self.train_output_info.values():
            m.reset()

    # ------------------------------------------------------------------ evaluation
    def eval(self, epoch_id: int = 0) -> Tuple[float, Dict[str, Dict[str, float]]]:
        """solver.py:684-711 + eval.py (_eval_by_dataset): returns (target metric, {validator: {metric.key: value}})."""
        if not self.validator:
            raise ValueError("Solver.eval needs at least one validator")
        target = float("inf")
        group: Dict[str, Dict[str, float]] = {}
        if self._is_operator:
            return self._eval_operator(epoch_id)
        for vname, val in self.validator.items():
            ds = getattr(val.data_loader, "dataset", val.data_loader)
            outs: Dict[str, list] = {}
            labs: Dict[str, list] = {}
            loss_sum: Dict[str, float] = {}
            nb = 0
            for (inp, lab, w) in val.data_loader:
                bsz = len(next(iter(inp.values())))
                key = (vname, bsz)
                if key not in self._compiled_val:
                    self._compiled_val[key] = CompiledConstraint(
                        vname, self.model, val.output_expr, list(ds.input_keys), list(ds.label_keys),
                        list((ds.weight or {}).keys()), val.loss, bsz, bsz, self.device, train=False, want_values=True,
                        extra_parameters=self._extra_parameters())
                cc = self._compiled_val[key]
                cc.bind(inp, lab, w)
                cc.fused.forward(self.model.materialize(), False)
                vals = cc.values()
                lv = cc.fused.losses()
                for k in cc.label_keys:
                    outs.setdefault(k, []).append(vals[k].clone())
                    labs.setdefault(k, []).append(torch.as_tensor(np.asarray(lab[k], dtype=np.float32)).to(self.device).view(-1, 1))
                    loss_sum[k] = loss_sum.get(k, 0.0) + lv[k]
                nb += 1
            n_total = len(ds) if hasattr(ds, "__len__") and not getattr(ds, "is_iterable", False) else None
            all_out = {k: self._gather_eval(torch.cat(v, 0), n_total) for k, v in outs.items()}
            all_lab = {k: self._gather_eval(torch.cat(v, 0), n_total) for k, v in labs.items()}
            group[vname] = {}
            for mname, metric in (val.metric or {}).items():
                res = metric(all_out, all_lab)
                for k, v in res.items():
                    group[vname][f"{mname}.{k}"] = _metric_value(v)
            msg = ", ".join(f"{k}: {v:.5f}" for k, v in group[vname].items())
            logger.info(f"[Eval][Epoch {epoch_id}][{vname}] loss: {sum(loss_sum.values()) / max(nb, 1):.5f}, {msg}")
            if group[vname] and target == float("inf"):
                target = float(next(iter(group[vname].values())))  # first metric of the first validator
        return target, group

    def _gather_eval(self, local: torch.Tensor, n_total: Optional[int]) -> torch.Tensor:
        """All ranks' shards in DATASET order, without the sampler's wrap-around padding (eval.py:154-161 truncates
        to num_samples): rank r holds samples r, r + W, r + 2W, ... of the padded index list."""
        if self.world_size == 1:
            return local
        full = misc.all_gather(local)  # rank-major: [rank 0's shard ; rank 1's shard ; ...]
        nl = local.shape[0]
        full = full.view(self.world_size, nl, *local.shape[1:]).transpose(0, 1).reshape(nl * self.world_size, *local.shape[1:])
        return full if n_total is None else full[:n_total]

    def _eval_operator(self, epoch_id: int):
        """eval.py _eval_by_dataset for models evaluated through torch (FNO): whole-dataset metrics."""
        from ..operator_engine import _to_dev

        target = float("inf")
        group: Dict[str, Dict[str, float]] = {}
        self.model.eval()
        for vname, val in self.validator.items():
            outs: Dict[str, list] = {}
            labs: Dict[str, list] = {}
            loss_sum, nb = 0.0, 0
            with torch.no_grad():
                for (inp, lab, w) in val.data_loader:
                    inp_d, lab_d, w_d = _to_dev(inp, self.device), _to_dev(lab, self.device), _to_dev(w, self.device)
                    out = self.model(inp_d)
                    data = {**inp_d, **out}
                    vals = {k: f(data) for k, f in val.output_expr.items()}
                    loss_sum += float(sum(val.loss(vals, lab_d, w_d).values()))
                    nb += 1
                    for k in vals:
                        outs.setdefault(k, []).append(vals[k])
                    for k in lab_d:
                        labs.setdefault(k, []).append(lab_d[k])
            ds = getattr(val.data_loader, "dataset", val.data_loader)
            n_total = len(ds) if hasattr(ds, "__len__") and not getattr(ds, "is_iterable", False) else None
            all_out = {k: self._gather_eval(torch.cat(v, 0), n_total) for k, v in outs.items()}
            all_lab = {k: self._gather_eval(torch.cat(v, 0), n_total) for k, v in labs.items()}
            group[vname] = {}
            for mname, metric in (val.metric or {}).items():
                for k, v in metric(all_out, all_lab).items():
                    group[vname][f"{mname}.{k}"] = _metric_value(v)
            msg = ", ".join(f"{k}: {v:.5f}" for k, v in group[vname].items())
            logger.info(f"[Eval][Epoch {epoch_id}][{vname}] loss: {loss_sum / max(nb, 1):.5f}, {msg}")
            if group[vname] and target == float("inf"):
                target = float(next(iter(group[vname].values())))
        self.model.train()
        return target, group

    # ------------------------------------------------------------------ prediction
    def predict(self, input_dict: Dict[str, Union[np.ndarray, torch.Tensor]], expr_dict: Optional[Dict[str, Callable]] = None,
                batch_size: Optional[int] = 64, no_grad: bool = True, return_numpy: bool = False):
        """solver.py:729-872.  With world_size > 1 the points are rank-strided (v[rank::world]) and the
        gathered result is restored to the input order, like the reference (:793-797, :847-855)."""
        if self._is_operator:
            from ..operator_engine import _to_dev

            n = len(next(iter(input_dict.values())))
            bs = n if batch_size is None else batch_size
            exprs = expr_dict if expr_dict is not None else {k: (lambda out, k=k: out[k]) for k in self.model.output_keys}
            res: Dict[str, list] = {k: [] for k in exprs}
            with torch.no_grad():
                for s0 in range(0, n, bs):
                    chunk = _to_dev({k: v[s0:s0 + bs] for k, v in input_dict.items()}, self.device)
                    data = {**chunk, **self.model(chunk)}
                    for k, f in exprs.items():
                        res[k].append(f(data))
            pred = {k: torch.cat(v, 0) for k, v in res.items()}
            return {k: v.cpu().numpy() for k, v in pred.items()} if return_numpy else pred
        if self._is_spinn:  # tensor-product grid of the three coordinate vectors (helmholtz3d.py:205-213)
            if expr_dict is not None:
                raise NotImplementedError("expr_dict with a SPINN model")
            out = self.model(input_dict)
            return {k: v.detach().cpu().numpy() for k, v in out.items()} if return_numpy else out
        n = len(next(iter(input_dict.values())))
        batch_size = n if batch_size is None else batch_size
        keys = list(input_dict.keys())
        arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).reshape(n, -1).astype(np.float32)
                for k, v in input_dict.items()}
        world, rank = self.world_size, self.rank
        pad = (-n) % world
        if pad:
            arrs = {k: np.concatenate([v, np.repeat(v[-1:], pad, 0)], 0) for k, v in arrs.items()}
        local = {k: v[rank::world] for k, v in arrs.items()} if world > 1 else arrs
        nl = len(next(iter(local.values())))
        exprs = expr_dict if expr_dict is not None else {k: (lambda out, k=k: out[k]) for k in self.model.output_keys}
        out_keys = list(exprs.keys())
        results = {k: [] for k in out_keys}
        for s in range(0, nl, batch_size):
            chunk = {k: v[s:s + batch_size] for k, v in local.items()}
            bsz = len(next(iter(chunk.values())))
            ck = (id(expr_dict), tuple(keys), bsz)
            if ck not in self._predict_cache:
                self._predict_cache[ck] = CompiledConstraint("predict", self.model, exprs, keys, [], [], None, bsz, bsz,
                                                            self.device, train=False, want_values=True,
                                                            extra_outputs=out_keys,
                                                            extra_parameters=self._extra_parameters())
            cc = self._predict_cache[ck]
            cc.bind(chunk, {}, {})
            cc.fused.forward(self.model.materialize(), False)
            vals = cc.values()
            for k in out_keys:
                results[k].append(vals[k].clone())
        pred = {k: torch.cat(v, 0) for k, v in results.items()}
        if world > 1:
            gathered = {k: misc.all_gather(v, concat=False) for k, v in pred.items()}
            pred = {}
            for k, parts in gathered.items():
                full = torch.empty((nl * world, 1), dtype=torch.float32, device=self.device)
                for r, p in enumerate(parts):
                    full[r::world] = p
                pred[k] = full[:n]
        if return_numpy:
            return {k: v.detach().cpu().numpy() for k, v in pred.items()}
        return pred

    def visualize(self, epoch_id: Optional[int] = None):
        """solver.py:713-727 + visu.py:32-101: every visualizer's expressions on its own points (the compiled forward of
        `predict`, in chunks of the visualizer's batch size), written by rank 0 under `<output_dir>/visual[/epoch_<n>]/`."""
        if not self.visualizer:
            raise ValueError("Solver.visualize needs at least one visualizer")
        for vis in self.visualizer.values():
            inputs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                      for k, v in vis.input_dict.items()}
            # visu.py:50-58 slices every column by the batch window of the FIRST one, so a column of another length (the "sdf"
            # that sample_initial_interior leaves in ldc2d_unsteady_Re10.py's hand-collated points) never reaches the model
            n = len(next(iter(inputs.values())))
            pred = self.predict({k: v for k, v in inputs.items() if len(v) == n}, vis.output_expr, batch_size=vis.batch_size,
                                return_numpy=True)
            if self.rank == 0:
                visual_dir = os.path.join(self.output_dir, "visual")
                if epoch_id:
                    visual_dir = os.path.join(visual_dir, f"epoch_{epoch_id}")
                os.makedirs(visual_dir, exist_ok=True)
                data = {k: np.asarray(v, dtype=np.float32) for k, v in inputs.items()}
                data.update({k: np.asarray(v, dtype=np.float32) for k, v in pred.items()})
                vis.save(os.path.join(visual_dir, vis.prefix), data)
        if isinstance(epoch_id, int):
            logger.info(f"[Visualize][Epoch {epoch_id}] Finish visualization")
        else:
            logger.info("[Visualize] Finish visualization")

    def plot_loss_history(self, by_epoch: bool = False, smooth_step: int = 1, use_semilogy: bool = True) -> None:
        """solver.py:1046-1076: the training-loss curves as `<output_dir>/<ylabel>.jpg`-style figure (misc.plot_curve).  The points
        are the iterations at which the loss was fetched from the device (every `log_freq`-th; the reference fetches every one)."""
        if not self.train_loss_info:
            logger.warning("plot_loss_history: no training loss has been recorded yet")
            return
        data = {}
        for key, hist in self.train_loss_info.items():
            if by_epoch:
                per_epoch: Dict[int, list] = {}
                for step, v in hist:
                    per_epoch.setdefault(step // max(self.iters_per_epoch, 1), []).append(v)
                data[key] = [float(np.mean(v)) for _, v in sorted(per_epoch.items())]
            else:
                data[key] = [v for _, v in hist]
        n = min(len(v) for v in data.values())
        misc.plot_curve({k: v[:n] for k, v in data.items()}, xlabel="Epoch" if by_epoch else "Iteration", ylabel="Loss",
                        output_dir=self.output_dir, smooth_step=smooth_step, use_semilogy=use_semilogy)

    @staticmethod
    def no_grad_context_manager(enable: bool):
        """solver.py:933-951.  The compiled forward never records a tape; the context only matters for user code on torch tensors."""
        import contextlib

        return torch.no_grad() if enable else contextlib.nullcontext()

    @staticmethod
    def autocast_context_manager(enable: bool, level: str = "O1"):
        """solver.py:913-931: AMP is not available on the fused path (fp32 only)."""
        import contextlib

        if enable:
            raise NotImplementedError("AMP is not available on the fused HIP path (fp32 only)")
        return contextlib.nullcontext()

    def export(self, *args, **kwargs):
        raise NotImplementedError("inference export (paddle.inference / ONNX) is out of scope")

    def finetune(self, pretrained_model_path: str) -> None:
        save_load.load_pretrain(self.model, pretrained_model_path, self.equation)
        self.train()
