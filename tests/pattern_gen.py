"""Random pattern generator shared by the CPU table-equality fuzz and the GPU result fuzz.
Grammar-directed so that most patterns parse and hit the reference's shape recognisers: literals,
classes, predefined classes, quantifiers, groups, alternations, anchors, plus some junk."""
import random

ATOMS = ["a", "b", "c", "x", "y", "z", "0", "1", "9", " ", "-", "@", "\\.", "\\d", "\\w", "\\s", ".",
         "[a-z]", "[0-9]", "[a-c]", "[A-Z]", "[abc]", "[^0-9]", "[^a-z]", "[a-z0-9]", "[a-zA-Z]", "[xyz]",
         "[0-9a-f]", "[ab]", "[.-]", "[\\s.-]", "[\\d]", "[a\\-z]", "[+]"]
QUANTS = ["", "", "", "", "+", "*", "?", "{2}", "{3}", "{1,3}", "{2,}", "{0,2}", "{1,}", "{4}"]
WORDS = ["foo", "bar", "baz", "hello", "cat", "dog", "ab", "abc", "xy", "http", "id", "no"]


def _atom(r):
    return r.choice(ATOMS)


def _seq(r, depth):
    n = r.choice([1, 1, 2, 2, 3, 4])
    out = []
    for _ in range(n):
        k = r.random()
        if k < 0.62 or depth >= 2:
            out.append(_atom(r) + r.choice(QUANTS))
        elif k < 0.75:
            out.append(r.choice(WORDS))
        elif k < 0.9:
            out.append("(" + _alt(r, depth + 1) + ")" + r.choice(QUANTS))
        else:
            out.append("(?:" + _alt(r, depth + 1) + ")" + r.choice(QUANTS))
    return "".join(out)


def _alt(r, depth):
    k = r.choice([1, 1, 1, 2, 2, 3, 4])
    parts = []
    for _ in range(k):
        parts.append(r.choice(WORDS) if r.random() < 0.4 else _seq(r, depth))
    return "|".join(parts)


def gen_pattern(r: random.Random) -> str:
    p = _alt(r, 0)
    k = r.random()
    if k < 0.08:
        p = "^" + p
    elif k < 0.14:
        p = p + "$"
    elif k < 0.2:
        p = "^" + p + "$"
    elif k < 0.23:   # junk: unbalanced / odd tokens the parser must reject (or accept) identically
        i = r.randrange(len(p) + 1)
        p = p[:i] + r.choice(["(", ")", "[", "]", "{", "}", "|", "\\", "*", "+", "?", "^", "$"]) + p[i:]
    return p


def patterns(seed: int, n: int):
    r = random.Random(seed)
    seen, out = set(), []
    while len(out) < n:
        p = gen_pattern(r)
        if p not in seen and len(p) <= 60:
            seen.add(p)
            out.append(p)
    return out
