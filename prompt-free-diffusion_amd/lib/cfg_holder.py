"""Process-wide configuration holder (same surface as the reference's lib/cfg_holder.py:15-28:
`cfg_unique_holder()` always returns the one instance; `.cfg`, `.code`, `.save_cfg`, `.add_code`)."""
import copy


class _CfgUniqueHolder:
    _instance = None

    def __init__(self):
        self.cfg = None
        self.code = set()  # names of the main code paths that have been entered

    def save_cfg(self, cfg):
        self.cfg = copy.deepcopy(cfg)

    def add_code(self, code):
        self.code.add(code)


def cfg_unique_holder(*args, **kwargs):
    if _CfgUniqueHolder._instance is None:
        _CfgUniqueHolder._instance = _CfgUniqueHolder(*args, **kwargs)
    return _CfgUniqueHolder._instance
