"""Logging helpers with the semantics of pymes/log.py:4-32 (indent = 4*level,
messages above ``debug_level`` are dropped)."""


def print_title(title_name, sep_symbol="=", level=1, debug_level=3):
    if level > debug_level:
        return
    level = max(level, 1)
    width = int(80 / level)
    if width < len(title_name):
        width = len(title_name) + 2
    shift = int((80 - width) / 2)
    pad = int((width - len(title_name)) / 2)
    print(" " * shift + sep_symbol * width)
    print(" " * (shift + pad) + title_name + " " * pad)
    print(" " * shift + sep_symbol * width)


def print_logging_info(*args, **kwargs):
    level = kwargs.get("level", 0)
    if level > kwargs.get("debug_level", 3):
        return
    print("    " * level + "".join(str(a) for a in args))
