"""String <-> id-sequence conversion (reference text/text_tools.py:9-73).

`text_to_onehot` cleans the text, drops every character that is not in the symbol table (and
the pad/eos symbols themselves), maps the rest to table indices and appends the eos id."""
import numpy as np

from . import cleaners
from .characters import chars, pad, eos

_char_to_ind = {c: i for i, c in enumerate(chars)}
_ind_to_char = dict(enumerate(chars))


def _should_keep(c):
    return c in _char_to_ind and c != pad and c != eos


def text_to_onehot(text, cleaner_name='basic_cleaners'):
    cleaner = getattr(cleaners, cleaner_name)
    ids = [_char_to_ind[c] for c in cleaner(text) if _should_keep(c)]
    ids.append(_char_to_ind[eos])
    return np.asarray(ids)


def onehot_to_text(onehot):
    return ''.join(_ind_to_char[int(o)] for o in onehot)
