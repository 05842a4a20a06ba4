"""`--normalize_text` (reference src/normalize_text.py, a character table adapted from ChemDataExtractor).

The table is third-party data that is not restated in this tree, and a partial one would change tokens silently;
the flag is therefore refused.  Every script defaults it to off, as the reference does."""


def normalize(text):
    raise NotImplementedError("--normalize_text: the reference's character table (ChemDataExtractor) is not rebuilt on this path")
