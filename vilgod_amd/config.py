"""A small Hydra/OmegaConf-compatible loader for the reference's config tree (hydra and omegaconf are not
installed in the target image; PyYAML is).

Implements exactly what `tools/preprocess_data.py` of the reference relies on (SURVEY §5 "Config / flags"):
  * a primary file with a `defaults:` list (`- group: file.yaml`, `- _self_`), group files under `<group>/`,
    nested `defaults` inside group files (`- base_cfg`), later entries override earlier ones;
  * command-line overrides `preprocessor=waymo` (group choice) and `a.b.c=value` (dotted assignment, YAML-typed);
  * `${a.b}` interpolation (absolute, or relative to the group's own package as Hydra places group files under
    the group key), resolved lazily so that overrides are honoured;
  * the resolvers registered at tools/preprocess_data.py:18-23: `as_tuple`, `join`, `format_split_join`,
    plus `now:` for the run directory;
  * `instantiate({_target_: pkg.mod.Class, ...})`.
Attribute and item access both work on the result (`cfg.preprocessor.clip.top_k`, `cfg['pipeline']`).
"""
import importlib
import os
import re
import time

import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _load_group_file(config_dir, group, name):
    name = name if name.endswith('.yaml') else name + '.yaml'
    path = os.path.join(config_dir, group, name)
    data = _load_yaml(path)
    out = {}
    defaults = data.pop('defaults', [])
    self_done = False
    for d in defaults:
        if d == '_self_':
            _merge(out, data)
            self_done = True
        elif isinstance(d, str):
            _merge(out, _load_group_file(config_dir, group, d))
        else:
            raise ValueError(f'{path}: unsupported defaults entry {d!r}')
    if not self_done:
        _merge(out, data)
    return out


_INTERP = re.compile(r'\$\{([^${}]+)\}')


def _lookup(root, dotted):
    cur = root
    for part in dotted.split('.'):
        if isinstance(cur, list):
            cur = cur[int(part)]
        else:
            cur = cur[part]
    return cur


def _resolve_value(root, value, package):
    if not isinstance(value, str) or '${' not in value:
        return value
    full = _INTERP.fullmatch(value)

    def one(expr):
        expr = expr.strip()
        if ':' in expr and not expr.startswith('oc.'):
            fn, arg = expr.split(':', 1)
            fn = fn.strip()
            if fn == 'now':
                return time.strftime(arg.strip())
            args = [_resolve_value(root, a.strip(), package) for a in _split_args(arg)]
            if fn == 'as_tuple':
                return tuple(yaml.safe_load(a) if isinstance(a, str) else a for a in args)
            if fn == 'join':
                return '_'.join(args[0])
            if fn == 'format_split_join':
                return '_'.join(str(args[0]).replace('\x00', '{').replace('\x01', '}').format('').split(' ')[:-1])
            raise KeyError(f'unknown resolver {fn}')
        for base in ([package] if package else []) + ['']:
            key = f'{base}.{expr}' if base else expr
            try:
                return _resolve_value(root, _lookup(root, key), package if base else _package_of(key))
            except (KeyError, IndexError, TypeError, ValueError):
                continue
        raise KeyError(f'interpolation ${{{expr}}} not found')

    if full:
        r = one(full.group(1))
        return r.replace('\x00', '{').replace('\x01', '}') if isinstance(r, str) else r
    while True:
        m = None
        for m in _INTERP.finditer(value):
            break
        if m is None:
            return value.replace('\x00', '{').replace('\x01', '}')
        # braces inside a substituted value must not be parsed as interpolation syntax of an enclosing resolver
        sub = str(one(m.group(1))).replace('{', '\x00').replace('}', '\x01')
        value = value[:m.start()] + sub + value[m.end():]


def _split_args(s):
    out, depth, cur = [], 0, ''
    for ch in s:
        if ch == ',' and depth == 0:
            out.append(cur)
            cur = ''
        else:
            depth += ch in '{['
            depth -= ch in '}]'
            cur += ch
    out.append(cur)
    return out


def _package_of(dotted):
    return dotted.split('.')[0] if '.' in dotted else ''


def _resolve_tree(root, node, package):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve_tree(root, node[k], package if package else (k if node is root else package))
        return node
    if isinstance(node, list):
        return [_resolve_tree(root, v, package) for v in node]
    return _resolve_value(root, node, package)


def load(config_dir, config_name='preprocessing', overrides=()):
    """-> Config.  overrides: iterable of 'key=value' strings (Hydra command-line syntax)."""
    primary = _load_yaml(os.path.join(config_dir, config_name if config_name.endswith('.yaml') else config_name + '.yaml'))
    defaults = primary.pop('defaults', ['_self_'])
    groups = []
    for d in defaults:
        if isinstance(d, dict):
            (g, f), = d.items()
            groups.append([g, f])
        else:
            groups.append([d, None])
    group_names = {g for g, f in groups if f is not None}
    dotted = []
    for ov in overrides:
        k, v = ov.split('=', 1)
        k = k.lstrip('+')
        if k in group_names:
            for gf in groups:
                if gf[0] == k:
                    gf[1] = v
        else:
            dotted.append((k, yaml.safe_load(v)))
    choice = {g: re.sub(r'\.yaml$', '', f) for g, f in groups if f is not None and '${' not in f}
    cfg = {}
    for g, f in groups:
        if f is None:
            if g == '_self_':
                _merge(cfg, primary)
            continue
        f = _INTERP.sub(lambda m: choice[m.group(1)], f)
        if g == 'hydra':
            continue                      # job logging / run dir: host concern, not part of the job config
        _merge(cfg, {g: _load_group_file(config_dir, g, f)})
    for k, v in dotted:
        cur = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            cur = cur[int(p)] if isinstance(cur, list) else cur.setdefault(p, {})
        if isinstance(cur, list):
            cur[int(parts[-1])] = v
        else:
            cur[parts[-1]] = v
    _resolve_tree(cfg, cfg, '')
    return _wrap(cfg)


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate for flat `_target_` dicts."""
    d = dict(cfg)
    target = d.pop('_target_')
    mod, attr = target.rsplit('.', 1)
    fn = getattr(importlib.import_module(mod), attr)
    d.update(kwargs)
    return fn(*args, **d)
