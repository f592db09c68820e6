"""Base class for custom evaluators -- host-side mirror of the reference's evaluators/evaluator_base.py:22-120."""
import copy
from collections import defaultdict
from functools import reduce
from pathlib import Path


class EvaluateBase(object):
    """Metric bookkeeping shared by evaluators (the reference keeps `_metric_values` on the class; one instance
    evaluates at a time there -- here it is per instance)."""

    def __init__(self):
        self._metric_values = defaultdict(list)

    @property
    def metric_values(self):
        return self._metric_values

    def append_metrics(self, pairs):
        for key, value in pairs.items():
            self._metric_values[key].append(value)

    def clear_metrics(self):
        for key in self._metric_values:
            self._metric_values[key].clear()

    def save_metrics(self, save_file, save_dir=None):
        max_len = reduce(max, [len(val) for val in self._metric_values.values()])
        temp_metrics = copy.deepcopy(self._metric_values)
        for key in self._metric_values:
            temp_metrics[key].extend(["--"] * (max_len - len(self._metric_values[key])))
        keys = list(temp_metrics.keys())
        save_path = Path(save_dir) / save_file if save_dir else Path(save_file)
        with save_path.open("w") as f:
            f.write(",".join(map(str, keys)) + "\n")
            for i in range(max_len):
                f.write(",".join([str(temp_metrics[key][i]) for key in keys]) + "\n")
        print("Write all metrics to", str(save_file))

    def run_with_session(self, session):
        raise NotImplementedError

    def run(self, input_fn, predict_keys=None, hooks=None, checkpoint_path=None):
        raise NotImplementedError

    def compare(self, cur_result, ori_result, primary_metric=None, secondary_metric=None):
        raise NotImplementedError
