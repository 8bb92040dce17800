"""Training statistics logger -- same interface as scene_synthesis/stats_logger.py (StatsLogger singleton with
running means, optional WandB subclass); ``wandb`` is imported lazily because it is an optional dependency."""
import sys


class AverageAggregator(object):
    def __init__(self):
        self._value = 0
        self._count = 0

    @property
    def value(self):
        return self._value / self._count

    @value.setter
    def value(self, val):
        self._value += val
        self._count += 1


class StatsLogger(object):
    __INSTANCE = None

    def __init__(self):
        if StatsLogger.__INSTANCE is not None:
            raise RuntimeError("StatsLogger should not be directly created")
        self._values = dict()
        self._loss = AverageAggregator()
        self._output_files = [sys.stdout]

    def add_output_file(self, f):
        self._output_files.append(f)

    def __getitem__(self, key):
        if key not in self._values:
            self._values[key] = AverageAggregator()
        return self._values[key]

    def clear(self):
        self._values.clear()
        self._loss = AverageAggregator()
        for f in self._output_files:
            if f.isatty():
                print(file=f, flush=True)

    def print_progress(self, epoch, batch, loss, precision="{:.5f}"):
        self._loss.value = loss
        msg = ("epoch: {} - batch: {} - loss: " + precision).format(epoch, batch, self._loss.value)
        for k, v in self._values.items():
            msg += " - " + k + ": " + precision.format(v.value)
        for f in self._output_files:
            if f.isatty():
                print(msg + "\b" * len(msg), end="", flush=True, file=f)
            else:
                print(msg, flush=True, file=f)

    @classmethod
    def instance(cls):
        if StatsLogger.__INSTANCE is None:
            StatsLogger.__INSTANCE = cls()
        return StatsLogger.__INSTANCE


class WandB(StatsLogger):
    """StatsLogger that also sends the running means to Weights & Biases when cleared (interface of the
    reference stats_logger.py:67-125: ``init(experiment_arguments, model, project, name, watch, log_frequency)``,
    validation values get a ``val_`` prefix when ``print_progress`` is called with a negative epoch)."""

    def init(self, experiment_arguments, model, project="experiment", name="experiment_name", watch=True,
             log_frequency=10):
        import wandb
        self.project, self.experiment_name = project, name
        self.watch, self.log_frequency = watch, log_frequency
        self._epoch, self._validation = 0, False
        wandb.login()
        wandb.init(project=(project or None), name=(name or None), config=dict(experiment_arguments.items()))
        if watch:
            wandb.watch(model, log_freq=log_frequency)

    def print_progress(self, epoch, batch, loss, precision="{:.5f}"):
        super().print_progress(epoch, batch, loss, precision)
        self._validation = epoch < 0
        if not self._validation:
            self._epoch = epoch

    def clear(self):
        import wandb
        prefix = "val_" if getattr(self, "_validation", False) else ""
        values = {prefix + k: v.value for k, v in self._values.items()}
        values[prefix + "loss"] = self._loss.value
        values[prefix + "epoch"] = getattr(self, "_epoch", 0)
        wandb.log(values)
        super().clear()
