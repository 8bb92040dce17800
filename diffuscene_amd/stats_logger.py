"""Running-mean statistics for the training loop, keyed by name.

Public surface used by the scripts and by train_on_batch / validate_on_batch (scene_synthesis/stats_logger.py):
``StatsLogger.instance()`` (process-wide singleton), ``logger[name].value = x`` (feed a sample; reading ``.value`` gives the
mean since the last ``clear()``), ``add_output_file``, ``print_progress(epoch, batch, loss)``, ``clear()``; ``WandB`` is the
same logger that also ships the means to Weights & Biases (``wandb`` is imported only when that class is used).
"""
import sys


class _RunningMean:
    """``m.value = x`` accumulates, ``m.value`` reads the mean of everything fed so far."""

    __slots__ = ("total", "n")

    def __init__(self):
        self.total, self.n = 0.0, 0

    def _get(self):
        return self.total / self.n

    def _feed(self, x):
        self.total += x
        self.n += 1

    value = property(_get, _feed)


AverageAggregator = _RunningMean          # name exported by the reference module


class StatsLogger:
    _singleton = None

    def __init__(self):
        if type(self)._shared() is not None:
            raise RuntimeError("StatsLogger should not be directly created")
        self._means = {}
        self._loss = _RunningMean()
        self._sinks = [sys.stdout]

    # -- singleton ------------------------------------------------------------------------------------------------
    @staticmethod
    def _shared():
        return StatsLogger._singleton

    @classmethod
    def instance(cls):
        if StatsLogger._singleton is None:
            StatsLogger._singleton = cls()
        return StatsLogger._singleton

    # -- feeding / reading ---------------------------------------------------------------------------------------
    def __getitem__(self, name):
        return self._means.setdefault(name, _RunningMean())

    def add_output_file(self, f):
        self._sinks.append(f)

    def _line(self, epoch, batch, precision):
        parts = ["epoch: %s" % epoch, "batch: %s" % batch, "loss: " + precision.format(self._loss.value)]
        parts += ["%s: %s" % (k, precision.format(m.value)) for k, m in self._means.items()]
        return " - ".join(parts)

    def print_progress(self, epoch, batch, loss, precision="{:.5f}"):
        self._loss.value = loss
        text = self._line(epoch, batch, precision)
        for sink in self._sinks:
            if sink.isatty():                       # terminals: rewrite the same line
                sink.write(text + "\b" * len(text))
            else:
                sink.write(text + "\n")
            sink.flush()

    def clear(self):
        self._means = {}
        self._loss = _RunningMean()
        for sink in self._sinks:
            if sink.isatty():
                sink.write("\n")
                sink.flush()


class WandB(StatsLogger):
    """Ships the means of every interval to Weights & Biases on ``clear()``; intervals whose ``print_progress`` calls used a
    negative epoch are validation intervals and get a ``val_`` prefix (reference stats_logger.py:67-125)."""

    def init(self, experiment_arguments, model, project="experiment", name="experiment_name", watch=True,
             log_frequency=10):
        import wandb
        self.project, self.experiment_name, self.watch, self.log_frequency = project, name, watch, log_frequency
        self._last_epoch, self._in_validation = 0, False
        wandb.login()
        wandb.init(project=project or None, name=name or None, config=dict(experiment_arguments.items()))
        if watch:
            wandb.watch(model, log_freq=log_frequency)

    def print_progress(self, epoch, batch, loss, precision="{:.5f}"):
        StatsLogger.print_progress(self, epoch, batch, loss, precision)
        self._in_validation = epoch < 0
        if epoch >= 0:
            self._last_epoch = epoch

    def clear(self):
        import wandb
        tag = "val_" if getattr(self, "_in_validation", False) else ""
        payload = {tag + k: m.value for k, m in self._means.items()}
        payload[tag + "loss"] = self._loss.value
        payload[tag + "epoch"] = getattr(self, "_last_epoch", 0)
        wandb.log(payload)
        StatsLogger.clear(self)
