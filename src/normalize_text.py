"""`--normalize_text` for the MI355X path: the character clean-up the reference applies to passages and queries before
tokenisation (its src/normalize_text.py, `normalize(text)`).  The rule set is ChemDataExtractor's
(chemdataextractor.text.normalize, Copyright 2016 by Matt Swain, MIT license: the copyright and permission notice are
kept in THIRD_PARTY_NOTICES.md at the repository root).

Behaviour, pinned code point by code point against the reference's own function (tests/golden/g14_text.json, made by
oracle/make_golden_text.py): C0 controls other than the white-space ones vanish, VT / FF / NEL become a space, every
hyphen / dash / minus variant becomes "-", the soft hyphen vanishes, curly and low double quotes become '"', single
quotes, apostrophes, grave / acute accents and (reversed) primes become one "'" per prime stroke, "..." replaces the
ellipsis character and a spaced-out " . . . ", fraction / division slashes become "/".  Tildes are left alone (the
reference has that rule commented out).  Built as ONE str.translate table instead of a chain of str.replace calls."""

_DASHES = "-\u2010\u2011\u2043\u2012\u2013\u2014\u2015\u2212\uff0d\u207b"       # hyphens, figure / en / em dashes, bar, minus signs
_DOUBLE_QUOTES = "\"\u201c\u201d\u201e\u201f"
_SINGLE_QUOTES = "'\u2018\u2019\u201a\u201b\u055a\ua78b\ua78c\uff07`\u00b4"        # quotes, apostrophes, grave and acute accent
_PRIME_STROKES = {0x2032: 1, 0x2035: 1, 0x2033: 2, 0x2036: 2, 0x2034: 3, 0x2037: 3, 0x2057: 4}
_SLASHES = "/\u2044\u2215"
_DROPPED_CONTROLS = [c for c in range(0x01, 0x1c) if c not in (0x09, 0x0a, 0x0b, 0x0c, 0x0d, 0x10)]
_TO_SPACE = (0x0b, 0x0c, 0x85)


def _build_table():
    table = {c: None for c in _DROPPED_CONTROLS}
    table[0xad] = None
    table.update({c: " " for c in _TO_SPACE})
    table.update({ord(c): "-" for c in _DASHES})
    table.update({ord(c): '"' for c in _DOUBLE_QUOTES})
    table.update({ord(c): "'" for c in _SINGLE_QUOTES})
    table.update({c: "'" * n for c, n in _PRIME_STROKES.items()})
    table.update({ord(c): "/" for c in _SLASHES})
    table[0x2026] = "..."
    return table


_TABLE = _build_table()


def normalize(text):
    return text.translate(_TABLE).replace(" . . . ", " ... ")
