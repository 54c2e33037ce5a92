"""Observation / action wrappers of the gym_jiminy pipeline for the vectorised, device-resident environments
(reference python/gym_jiminy/common/gym_jiminy/common/wrappers/: observation_stack.py, normalize.py,
flatten.py).  They wrap anything with the `reset / step / observation` surface of `VecJiminyEnv`; observations are
nested dicts of tensors with a leading batch axis, and stay on the device.

* `StackObservation(env, num_stack, nested_filter_keys, skip_frames_ratio)`: rolling stack of the selected leaves;
  a stacked leaf `(B, ...)` becomes `(B, num_stack, ...)`, oldest first, the latest frame always the current value
  (observation_stack.py:40-265).  Lanes that were re-initialised restart with a zero history.
* `NormalizeObservation(env, bounds)` / `NormalizeAction(env, low, high)`: affine map of bounded leaves to [-1, 1]
  from pre-defined bounds, without clipping (normalize.py:49-114); unbounded leaves are left as they are
  (`ignore_unbounded`).
* `FlattenObservation(env)`: all leaves concatenated into one `(B, D)` tensor, keys in sorted order (flatten.py).
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch

Path = Tuple[Union[str, int], ...]


def flatten_with_path(tree: Any, prefix: Path = ()) -> List[Tuple[Path, torch.Tensor]]:
    """Leaves of a nested dict / list / tuple of tensors with their paths, dict keys in sorted order."""
    if isinstance(tree, dict):
        out: List[Tuple[Path, torch.Tensor]] = []
        for k in sorted(tree):
            out += flatten_with_path(tree[k], prefix + (k,))
        return out
    if isinstance(tree, (list, tuple)):
        out = []
        for i, v in enumerate(tree):
            out += flatten_with_path(v, prefix + (i,))
        return out
    return [(prefix, tree)]


def _set_path(tree: Any, path: Path, value: Any) -> Any:
    """Copy-on-write update of one leaf of a nested dict."""
    if not path:
        return value
    new = dict(tree)
    new[path[0]] = _set_path(tree[path[0]], path[1:], value)
    return new


class _Wrapper:
    def __init__(self, env: Any) -> None:
        self.env = env

    def __getattr__(self, name: str) -> Any:   # everything else is the wrapped environment's
        return getattr(self.env, name)

    def transform(self, obs: Any) -> Any:
        return obs

    def observation(self) -> Any:
        """Current observation.  Read-only with respect to the wrapper's own state: wrappers that accumulate (the
        observation stack) update in `reset` / `step` only and answer here from what they hold (`peek`)."""
        return self.peek(self.env.observation())

    def peek(self, obs: Any) -> Any:
        return self.transform(obs)

    def reset(self, *args: Any, **kw: Any):
        obs, info = self.env.reset(*args, **kw)
        self._on_reset(None)
        return self.transform(obs), info

    def step(self, action: torch.Tensor):
        obs, reward, terminated, truncated, info = self.env.step(self.transform_action(action))
        if "reset_mask" in info:
            self._on_reset(info["reset_mask"])
        return self.transform(obs), reward, terminated, truncated, info

    def transform_action(self, action: torch.Tensor) -> torch.Tensor:
        return action

    def _on_reset(self, lane_mask: Optional[torch.Tensor]) -> None:
        pass


class StackObservation(_Wrapper):
    def __init__(self, env: Any, *, num_stack: int, nested_filter_keys: Optional[Sequence[Union[Path, str]]] = None,
                 skip_frames_ratio: int = -1) -> None:
        super().__init__(env)
        if num_stack < 1:
            raise ValueError("num_stack must be at least 1")
        self.num_stack = int(num_stack)
        # -1: one update per environment step (the only refresh granularity of the vectorised environments);
        # n >= 0: n steps skipped between two shifts of the stack
        self.skip_frames_ratio = int(skip_frames_ratio)
        keys = [(k,) if isinstance(k, (str, int)) else tuple(k) for k in (nested_filter_keys or [()])]
        self.nested_filter_keys: List[Path] = keys
        self._stack: Dict[Path, torch.Tensor] = {}
        self._n_since_shift = 0

    def _selected(self, path: Path) -> bool:
        return any(path[:len(k)] == k for k in self.nested_filter_keys)

    def _on_reset(self, lane_mask: Optional[torch.Tensor]) -> None:
        if lane_mask is None:
            self._stack.clear()
            self._n_since_shift = 0
            return
        for t in self._stack.values():
            m = lane_mask.view(-1, *([1] * (t.dim() - 1)))
            t.copy_(torch.where(m, torch.zeros_like(t), t))

    def transform(self, obs: Any) -> Any:
        leaves = [(p, x) for p, x in flatten_with_path(obs) if self._selected(p)]
        if not leaves:
            raise ValueError("At least one observation leaf must be stacked.")
        shift = self.skip_frames_ratio < 0 or self._n_since_shift >= self.skip_frames_ratio
        out = obs
        for path, x in leaves:
            st = self._stack.get(path)
            if st is None:
                st = torch.zeros((x.shape[0], self.num_stack) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                self._stack[path] = st
            elif shift and self.num_stack > 1:
                st[:, :-1] = st[:, 1:].clone()
            st[:, -1] = x
            # (a copy: the ring buffer is rewritten by the next step, a rollout buffer that keeps the returned
            # observation must not see that)
            out = _set_path(out, path, st.clone())
        self._n_since_shift = 0 if shift else self._n_since_shift + 1
        return out

    def peek(self, obs: Any) -> Any:
        """The stack as it stands (copies), without shifting it: `observation()` between two steps."""
        out = obs
        for path, x in flatten_with_path(obs):
            if self._selected(path):
                st = self._stack.get(path)
                if st is None:   # nothing stacked yet: zeros and the current frame, like the first `transform`
                    st = torch.zeros((x.shape[0], self.num_stack) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                    st[:, -1] = x
                out = _set_path(out, path, st.clone())
        return out


class NormalizeObservation(_Wrapper):
    def __init__(self, env: Any, bounds: Dict[Path, Tuple[Any, Any]], ignore_unbounded: bool = True) -> None:
        """`bounds`: `{path: (low, high)}` per leaf (broadcastable to the leaf without its batch axis)."""
        super().__init__(env)
        self.ignore_unbounded = ignore_unbounded
        self._bounds = {((p,) if isinstance(p, (str, int)) else tuple(p)): b for p, b in bounds.items()}
        self._maps: Dict[Path, Tuple[torch.Tensor, torch.Tensor]] = {}

    def _map(self, path: Path, x: torch.Tensor) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        if path in self._maps:
            return self._maps[path]
        if path not in self._bounds:
            if self.ignore_unbounded:
                return None
            raise ValueError(f"no bounds for observation leaf {path}")
        lo = torch.as_tensor(self._bounds[path][0], dtype=x.dtype, device=x.device).expand(x.shape[1:]).clone()
        hi = torch.as_tensor(self._bounds[path][1], dtype=x.dtype, device=x.device).expand(x.shape[1:]).clone()
        bounded = torch.isfinite(lo) & torch.isfinite(hi)
        if not bool(bounded.all()) and not self.ignore_unbounded:
            raise ValueError(f"observation leaf {path} is not bounded")
        # (x - mean) / scale with mean = (hi + lo) / 2, scale = (hi - lo) / 2; identity where unbounded
        mean = torch.where(bounded, 0.5 * (hi + lo), torch.zeros_like(lo))
        scale = torch.where(bounded, 0.5 * (hi - lo), torch.ones_like(lo))
        self._maps[path] = (mean, scale)
        return self._maps[path]

    def transform(self, obs: Any) -> Any:
        out = obs
        for path, x in flatten_with_path(obs):
            m = self._map(path, x)
            if m is not None:
                out = _set_path(out, path, (x - m[0]) / m[1])
        return out


class NormalizeAction(_Wrapper):
    def __init__(self, env: Any, low: Any, high: Any) -> None:
        """The policy acts in [-1, 1]^M; the environment receives `mean + scale * action` (no clipping)."""
        super().__init__(env)
        self._low, self._high = low, high
        self._map: Optional[Tuple[torch.Tensor, torch.Tensor]] = None

    def transform_action(self, action: torch.Tensor) -> torch.Tensor:
        if self._map is None:
            lo = torch.as_tensor(self._low, dtype=action.dtype, device=action.device).expand(action.shape[1:])
            hi = torch.as_tensor(self._high, dtype=action.dtype, device=action.device).expand(action.shape[1:])
            bounded = torch.isfinite(lo) & torch.isfinite(hi)
            self._map = (torch.where(bounded, 0.5 * (hi + lo), torch.zeros_like(lo)),
                         torch.where(bounded, 0.5 * (hi - lo), torch.ones_like(lo)))
        return self._map[0] + self._map[1] * action


class FlattenObservation(_Wrapper):
    def __init__(self, env: Any, dtype: Optional[torch.dtype] = None, exclude: Iterable[Path] = (("t",),)) -> None:
        super().__init__(env)
        self._dtype = dtype
        self._exclude = [tuple(e) for e in exclude]

    def transform(self, obs: Any) -> torch.Tensor:
        parts = [x.reshape(x.shape[0], -1) for p, x in flatten_with_path(obs)
                 if not any(p[:len(e)] == e for e in self._exclude)]
        flat = torch.cat(parts, dim=1)
        return flat.to(self._dtype) if self._dtype is not None else flat
