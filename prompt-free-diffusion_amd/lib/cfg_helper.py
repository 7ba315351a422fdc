"""YAML model-config bank with the semantics of the reference's lib/cfg_helper.py:21-171:

* `model_cfg_bank()(name)` -> attribute-dict with `.type/.args/.name/.symbol/...`
* the yaml file is chosen by the name's prefix (cfg_helper.py:148-171)
* `super_cfg: parent` inherits every field of `parent`; `args` are MERGED (child wins per key),
  every other field is replaced (cfg_helper.py:121-133); `delete_args` drops inherited args
* string values `MODEL(x)` are replaced by the resolved config `x`; `SAME(a.b)` by the value at
  path a.b of the config root; `SEARCH(a.b)` by a depth-first search for that path
  (cfg_helper.py:21-100)
* known-broken entries of the reference yaml stay broken on purpose (`seecoder_pa: super_cfg:
  seet`, `pdf_seecoder_pa`) so that both trees fail the same way.

Only the model bank is on the hot path; the dataset bank / experiment CLI of the reference
(cfg_helper.py:173-666) belong to its dead training scaffold and are out of scope.
Host-side Python: no kernels here.
"""
import copy
import os
import os.path as osp

import yaml


class AttrDict(dict):
    """dict with attribute access, nested dicts (also inside lists/tuples) wrapped recursively;
    the subset of easydict.EasyDict the reference relies on."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = {} if d is None else dict(d)
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kwargs):
        d = {} if other is None else dict(other)
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


edict = AttrDict  # the name the reference imports

_PREFIX_TO_FILE = (
    ("openai_unet", "openai_unet.yaml"),
    ("clip", "clip.yaml"),
    ("autokl", "autokl.yaml"),
    ("controlnet", "controlnet.yaml"),
    ("swin", "swin.yaml"),
    ("pfd", "pfd.yaml"),
    ("seecoder", "seecoder.yaml"),
)


def _walk_path(root, dotted):
    node = root
    for part in (p.strip() for p in dotted.split(".")):
        try:
            part = int(part)
        except ValueError:
            pass
        node = node[part]
    return node


def cfg_solvef(cmd, root):
    """resolve one scalar config value (SAME / SEARCH / MODEL directives)"""
    if not isinstance(cmd, str):
        return cmd
    if cmd.startswith("SAME"):
        try:
            return cfg_solvef(_walk_path(root, cmd[4:].strip("()")), root)
        except (KeyError, IndexError, TypeError):
            return cmd
    if cmd.startswith("SEARCH"):
        try:
            return cfg_solvef(_walk_path(root, cmd[6:].strip("()")), root)
        except (KeyError, IndexError, TypeError):
            children = root.values() if isinstance(root, dict) else root if isinstance(root, list) else ()
            for child in children:
                if isinstance(child, (dict, list)):
                    rv = cfg_solvef(cmd, child)
                    if rv != cmd:
                        return rv
            return cmd
    if cmd.startswith("MODEL"):
        return model_cfg_bank()(cmd[5:].strip("()"))
    if cmd.startswith("DATASET"):
        raise NotImplementedError("dataset configs are not part of the inference hot path")
    return cmd


def cfg_solve(cfg, cfg_root):
    """resolve every directive inside a nested dict/list config, in place"""
    items = enumerate(cfg) if isinstance(cfg, list) else cfg.items() if isinstance(cfg, dict) else ()
    for k, v in list(items):
        if isinstance(v, tuple):
            v = list(v)
        cfg[k] = cfg_solve(v, cfg_root) if isinstance(v, (list, dict)) else cfg_solvef(v, cfg_root)
    return cfg


class model_cfg_bank(object):
    def __init__(self, cfg_dir=None):
        if cfg_dir is None:
            cwd_dir = osp.join("configs", "model")  # the reference resolves relative to the CWD
            pkg_dir = osp.normpath(osp.join(osp.dirname(osp.abspath(__file__)), "..", "configs", "model"))
            cfg_dir = os.environ.get("PFD_CFG_DIR") or (cwd_dir if osp.isdir(cwd_dir) else pkg_dir)
        self.cfg_dir = cfg_dir
        self.cfg_bank = AttrDict()

    def get_yaml_path(self, name):
        for prefix, fname in _PREFIX_TO_FILE:
            if name.startswith(prefix):
                return osp.join(self.cfg_dir, fname)
        raise ValueError(name)

    def __call__(self, name):
        if name not in self.cfg_bank:
            with open(self.get_yaml_path(name), "r") as f:
                self.cfg_bank.update(AttrDict(yaml.load(f, Loader=yaml.FullLoader)))
        cfg = self.cfg_bank[name]
        cfg.name = name
        if "super_cfg" in cfg:
            merged = self(cfg.super_cfg)  # a deep copy of the resolved parent
            if "args" in cfg:
                if "args" in merged:
                    merged.args.update(cfg.args)
                else:
                    merged.args = cfg.args
                cfg.pop("args")
            merged.update(cfg)
            merged.pop("super_cfg")
            for dropped in merged.pop("delete_args", []):
                merged.args.pop(dropped)
            cfg = merged
        cfg = cfg_solve(cfg, cfg)
        self.cfg_bank[name] = cfg
        return copy.deepcopy(cfg)
