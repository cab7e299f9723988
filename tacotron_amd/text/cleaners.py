"""Text cleaners (reference text/cleaners.py:13-70). `basic_cleaners` is the one the synthesis
path uses (synthesizer.py:38). `transliteration_cleaners` needs the optional `unidecode`
package and raises a clear error without it."""
import re

_whitespace_re = re.compile(r'\s+')


def lowercase(text):
    return text.lower()


def remove_whitespace(text):
    return _whitespace_re.sub(' ', text)


def transliterate(text):
    try:
        from unidecode import unidecode
    except ImportError as e:  # pragma: no cover - optional dependency
        raise RuntimeError('transliterate() needs the `unidecode` package') from e
    return unidecode(text)


def basic_cleaners(text):
    return remove_whitespace(lowercase(text))


def transliteration_cleaners(text):
    return remove_whitespace(lowercase(transliterate(text)))
