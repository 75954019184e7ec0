"""Minimal stand-in for ``tf.flags``: a global FLAGS namespace populated at import time by DEFINE_* calls
scattered over the modules, exactly like the reference (W/train.py:38-137, W/frame_level_models.py:20-84,
W/video_level_models.py:19-47, W/losses.py:22-44).  Flag NAMES and DEFAULTS are the reference's so its shell
scripts remain a valid source of configurations.
"""


class _Flags(object):
    def __init__(self):
        object.__setattr__(self, "_defs", {})

    def _define(self, name, default, help_, kind):
        defs = object.__getattribute__(self, "_defs")
        if name in defs:  # re-import / duplicate definition keeps the current value (tf.flags raises; we tolerate)
            return
        defs[name] = {"default": default, "value": default, "help": help_, "kind": kind}

    def __getattr__(self, name):
        defs = object.__getattribute__(self, "_defs")
        if name not in defs:
            raise AttributeError("unknown flag %r" % name)
        return defs[name]["value"]

    def __setattr__(self, name, value):
        defs = object.__getattribute__(self, "_defs")
        if name not in defs:
            raise AttributeError("unknown flag %r" % name)
        defs[name]["value"] = value

    def __contains__(self, name):
        return name in object.__getattribute__(self, "_defs")

    def reset(self):
        for d in object.__getattribute__(self, "_defs").values():
            d["value"] = d["default"]

    def parse(self, argv):
        """--name=value / --name value / --boolflag / --noboolflag."""
        defs = object.__getattribute__(self, "_defs")
        i, rest = 0, []
        while i < len(argv):
            a = argv[i]
            if not a.startswith("--"):
                rest.append(a); i += 1; continue
            body = a[2:]
            if "=" in body:
                name, val = body.split("=", 1)
            else:
                name, val = body, None
            if name not in defs and name.startswith("no") and name[2:] in defs and defs[name[2:]]["kind"] is bool:
                defs[name[2:]]["value"] = False; i += 1; continue
            if name not in defs:
                raise ValueError("unknown flag --%s" % name)
            kind = defs[name]["kind"]
            if val is None:
                if kind is bool:
                    val = "true"
                else:
                    i += 1
                    val = argv[i]
            if kind is bool:
                defs[name]["value"] = str(val).lower() in ("1", "true", "t", "yes", "y")
            else:
                defs[name]["value"] = kind(val)
            i += 1
        return rest

    def as_dict(self):
        return {k: v["value"] for k, v in object.__getattribute__(self, "_defs").items()}


FLAGS = _Flags()


def DEFINE_integer(name, default, help_=""):
    FLAGS._define(name, default, help_, int)


def DEFINE_float(name, default, help_=""):
    FLAGS._define(name, default, help_, float)


def DEFINE_string(name, default, help_=""):
    FLAGS._define(name, default, help_, str)


def DEFINE_bool(name, default, help_=""):
    FLAGS._define(name, default, help_, bool)


DEFINE_boolean = DEFINE_bool
