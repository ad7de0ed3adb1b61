"""detectron2.utils.events plumbing: EventStorage context + get_event_storage().  Scalar names
logged by the reference are preserved by the model (metrics row)."""
from collections import defaultdict

_CURRENT_STORAGE_STACK = []


def get_event_storage():
    assert len(_CURRENT_STORAGE_STACK), "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT_STORAGE_STACK[-1]


def has_event_storage():
    return len(_CURRENT_STORAGE_STACK) > 0


class EventStorage:
    def __init__(self, start_iter=0):
        self._history = defaultdict(list)
        self._latest = {}
        self._iter = start_iter
        self._vis_data = []

    def put_image(self, img_name, img_tensor):
        self._vis_data.append((img_name, img_tensor, self._iter))

    def put_scalar(self, name, value, smoothing_hint=True, cur_iter=None):
        value = float(value)
        it = self._iter if cur_iter is None else cur_iter
        self._history[name].append((value, it))
        self._latest[name] = (value, it)

    def put_scalars(self, *, smoothing_hint=True, cur_iter=None, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint=smoothing_hint, cur_iter=cur_iter)

    def history(self, name):
        if name not in self._history:
            raise KeyError(f"No history metric available for {name}!")
        return self._history[name]

    def histories(self):
        return self._history

    def latest(self):
        return self._latest

    def step(self):
        self._iter += 1

    @property
    def iter(self):
        return self._iter

    @iter.setter
    def iter(self, val):
        self._iter = int(val)

    def clear_images(self):
        self._vis_data = []

    def __enter__(self):
        _CURRENT_STORAGE_STACK.append(self)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        assert _CURRENT_STORAGE_STACK[-1] == self
        _CURRENT_STORAGE_STACK.pop()
