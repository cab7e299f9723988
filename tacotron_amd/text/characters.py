"""Symbol table of the text frontend (reference text/characters.py:1-12): 95 symbols,
pad id 0, eos id 1, then A-Z, a-z, punctuation + space, 20 Icelandic letters, digits."""
pad = '_'
eos = '~'

_latin = [chr(c) for c in range(ord('A'), ord('Z') + 1)] + [chr(c) for c in range(ord('a'), ord('z') + 1)]
_punct = list("!'(),-.:;? ")
_isk_upper = list('ÁÐÉÍÓÚÝÞÆÖ')
_isk_lower = [c.lower() for c in _isk_upper]
_digits = [str(d) for d in range(10)]

chars = [pad, eos] + _latin + _punct + _isk_upper + _isk_lower + _digits
