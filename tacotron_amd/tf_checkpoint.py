"""TF V2 checkpoint ("tensor bundle") reader and writer, no TensorFlow needed.

Replaces `tf.train.Saver().restore(session, checkpoint_path)` of /root/reference/synthesizer.py:32-34 (and the
save side of model/tacotron.py:133,169) for the variables of the inference graph: reads
`<prefix>.index` + `<prefix>.data-0000N-of-0000M` as TF 1.15 writes them (format: SURVEY.md Appendix B, restated
from the published TF / LevelDB sources; TF itself is not installed here and the reference ships no checkpoint,
so this module is checked against its own writer only - "parity unpinned", tests/test_cpu_tf_checkpoint.py).
Variables are matched to the graph by pattern and shape (match_variables), tensor CRCs are verified by default.

  index file  = LevelDB-style immutable table: data blocks of prefix-compressed entries
                (varint shared, varint non_shared, varint value_len, key suffix, value) + restart array + count,
                each block followed by a compression byte (0 = none, the bundle writer's setting; 1 = snappy is
                rejected) and a masked CRC32C; an index block mapping separator keys to block handles; a 48-byte
                footer (metaindex handle, index handle, padding, magic 0xdb4775248b80fb57).
  key ""      -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
  key <name>  -> BundleEntryProto {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id, 4: offset,
                                   5: size, 6: crc32c (fixed32, masked)}
  data files  = raw little-endian row-major tensor bytes at [offset, offset + size).
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}      # DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ---- CRC32C (Castagnoli), masked as LevelDB / TF store it ---------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def _crc_scalar(c, data):
    """Raw register update (no pre/post inversion) over `data`, one byte at a time."""
    t = _crc_table()
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


def crc32c(data, crc=0):
    """CRC-32C of `data` (continuing from `crc`). Large inputs are cut into equal chunks whose registers advance in
    lock-step as one numpy vector (a table lookup per byte position for all chunks at once); the chunk results are
    then chained with the CRC's linearity: state(c, d) = advance_len(d)(c) ^ state(0, d)."""
    data = bytes(data)
    n = len(data)
    c = crc ^ 0xFFFFFFFF
    if n < 1 << 14:
        return _crc_scalar(c, data) ^ 0xFFFFFFFF
    lanes = min(1 << 16, n // 64)
    L = n // lanes
    head = n - lanes * L
    c = _crc_scalar(c, data[:head])
    t = np.asarray(_crc_table(), np.uint32)
    body = np.frombuffer(data, np.uint8, count=lanes * L, offset=head).reshape(lanes, L)
    st = np.zeros(lanes, np.uint32)
    # Z[k][v]: the register (v << 8k) advanced over L zero bytes
    z = (np.arange(256, dtype=np.uint32)[None, :] << (8 * np.arange(4, dtype=np.uint32))[:, None]).reshape(-1)
    for j in range(L):
        st = t[(st ^ body[:, j]) & 0xFF] ^ (st >> 8)
        z = t[z & 0xFF] ^ (z >> 8)
    z = z.reshape(4, 256).tolist()
    z0, z1, z2, z3 = z
    for s_i in st.tolist():
        c = z0[c & 0xFF] ^ z1[(c >> 8) & 0xFF] ^ z2[(c >> 16) & 0xFF] ^ z3[c >> 24] ^ s_i
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / minimal protobuf -----------------------------------------------------------------------------
def _get_varint(buf, pos):
    out, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise CheckpointError('varint too long')


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """{field: [values]} with varints as int, fixed32/64 as int, length-delimited as bytes."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise CheckpointError('truncated protobuf field')
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError('unsupported protobuf wire type %d' % wt)
        out.setdefault(f, []).append(v)
    return out


def _field(f, wt, payload):
    return _put_varint((f << 3) | wt) + payload


# ---- table reading ------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
    block = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if len(block) != size or len(trailer) != 5:
        raise CheckpointError('block handle (%d, %d) outside the index file' % (offset, size))
    if trailer[0] != 0:
        raise CheckpointError('compressed table block (type %d): only uncompressed bundles are supported' % trailer[0])
    if verify:
        want = struct.unpack('<I', trailer[1:5])[0]
        got = mask_crc(crc32c(bytes(block) + bytes(trailer[:1])))
        if want != got:
            raise CheckpointError('index file block at %d fails its CRC32C' % offset)
    return block


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError('table block too short')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    if end < 0:
        raise CheckpointError('bad restart array')
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise CheckpointError('corrupt table entry')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_index(prefix, verify=True):
    """{name: dict(dtype, shape, shard_id, offset, size, crc32c)}, header dict."""
    path = prefix + '.index'
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 48:
        raise CheckpointError('%s: shorter than a table footer' % path)
    footer = buf[-48:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError('%s: not a TF tensor-bundle index (bad magic)' % path)
    pos = 0
    _, pos = _get_varint(footer, pos)       # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries, header = {}, None
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, value in _block_entries(_read_block(buf, boff, bsize, verify)):
            msg = _parse_proto(value)
            if key == b'':
                header = dict(num_shards=msg.get(1, [1])[0], endianness=msg.get(2, [0])[0])
                continue
            shape = []
            if 2 in msg:
                for dim in _parse_proto(msg[2][0]).get(2, []):
                    shape.append(_parse_proto(dim).get(1, [0])[0])
            if 7 in msg:
                raise CheckpointError('%s: sliced (partitioned) variables are not supported' % key.decode())
            entries[key.decode('utf-8')] = dict(dtype=msg.get(1, [0])[0], shape=tuple(shape),
                                                shard_id=msg.get(3, [0])[0], offset=msg.get(4, [0])[0],
                                                size=msg.get(5, [0])[0], crc32c=msg.get(6, [None])[0])
    if header is None:
        raise CheckpointError('%s: no bundle header entry' % path)
    if header['endianness'] != 0:
        raise CheckpointError('%s: big-endian bundle' % path)
    return entries, header


def read_checkpoint(prefix, names=None, verify_crc=True):
    """{name: ndarray} of the bundle's tensors (all, or only `names`). verify_crc (default) checks every tensor's
    CRC32C against its index entry (vectorised, ~0.2 s for the 28 MB of a Tacotron checkpoint); the index blocks are
    always checked."""
    entries, header = read_index(prefix)
    out, files = {}, {}
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e['dtype'] not in _DTYPES:
                raise CheckpointError('%s: unsupported dtype enum %d' % (name, e['dtype']))
            dt = np.dtype(_DTYPES[e['dtype']])
            count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
            if count * dt.itemsize != e['size']:
                raise CheckpointError('%s: %d bytes stored for shape %s of %s' % (name, e['size'], e['shape'], dt))
            sid = e['shard_id']
            if sid not in files:
                files[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards']), 'rb')
            f = files[sid]
            f.seek(e['offset'])
            raw = f.read(e['size'])
            if len(raw) != e['size']:
                raise CheckpointError('%s: data shard %d is truncated' % (name, sid))
            if verify_crc and e['crc32c'] is not None and mask_crc(crc32c(raw)) != e['crc32c']:
                raise CheckpointError('%s: tensor bytes fail their CRC32C' % name)
            out[name] = np.frombuffer(raw, dtype=dt.newbyteorder('<')).astype(dt).reshape(e['shape'])
    finally:
        for f in files.values():
            f.close()
    return out


# ---- writing (the save side; also what the reader is tested against) -----------------------------------------
def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        out += key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_checkpoint(prefix, tensors, block_bytes=4096):
    """Write {name: ndarray} as a one-shard bundle (sorted keys, uncompressed blocks, CRCs filled in)."""
    names = sorted(tensors, key=lambda n: n.encode('utf-8'))
    data = bytearray()
    items = [(b'', _field(1, 0, _put_varint(1)) + _field(2, 0, _put_varint(0)) +
              _field(3, 2, _put_varint(2) + _field(1, 0, _put_varint(1))))]     # version {producer: 1}
    for n in names:
        a = np.asarray(tensors[n], order='C')
        if a.dtype not in _DTYPE_IDS:
            raise CheckpointError('%s: dtype %s cannot be stored' % (n, a.dtype))
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        shape = b''.join(_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_field(1, 0, _put_varint(int(s)))))
                         for s in a.shape)
        msg = _field(1, 0, _put_varint(_DTYPE_IDS[a.dtype])) + _field(2, 2, _put_varint(len(shape)) + shape)
        if len(data):
            msg += _field(4, 0, _put_varint(len(data)))
        msg += _field(5, 0, _put_varint(len(raw))) + _field(6, 5, struct.pack('<I', mask_crc(crc32c(raw))))
        items.append((n.encode('utf-8'), msg))
        data += raw
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.extend(b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return _put_varint(off) + _put_varint(len(block))

    index_items, cur, cur_bytes = [], [], 0
    for kv in items:
        cur.append(kv)
        cur_bytes += len(kv[0]) + len(kv[1]) + 3
        if cur_bytes >= block_bytes:
            index_items.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index_items.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, restart_interval=1))
    footer = meta + index
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
    with open('%s.data-00000-of-00001' % prefix, 'wb') as f:
        f.write(bytes(data))


# ---- the reference's variables ---------------------------------------------------------------------------------
GRAPH_PREFIX = 'model/inference/'      # scopes of synthesizer.py:23 and model/tacotron.py:31


_BN_RE = re.compile(r'^(.*?)batch_normalization(?:_(\d+))?/(gamma|beta|moving_mean|moving_variance)$')
# scope segments some TF 1.x releases insert and others do not (`cell_{1,2}/[residual_wrapper/]
# gru_cell/...`, model/decoder.py:28-32): dropped from both sides before names are compared
_OPTIONAL_SEGMENTS = ('residual_wrapper/',)


def _canonical(rel):
    for seg in _OPTIONAL_SEGMENTS:
        rel = rel.replace(seg, '')
    return rel


def match_variables(entries, hp=None):
    """Match the variables of a checkpoint index ({full name: {'shape': ...}}) to the inference graph's variables by
    pattern and shape, not by exact string:
      * names are taken relative to `model/inference/`; optimiser slots (`.../Adam`, `.../Adam_1`), `global_step` and
        everything outside that scope are ignored;
      * optional scope segments (`residual_wrapper/`) may be present or absent;
      * batch-norm scopes are matched by ORDER inside their enclosing scope: `tf.layers.batch_normalization` is called
        outside the conv's own scope (model/modules.py:138-140), so TF auto-numbers it (`batch_normalization`,
        `_1`, ...), per enclosing scope in TF 1.15 but graph-wide under other naming rules; the k-th expected BN of a
        scope takes the k-th present one, sorted by its number;
      * a shape mismatch is never papered over.
    Returns (mapping {graph name: checkpoint name}, problems [str], leftovers [checkpoint names under the scope that
    nothing claimed])."""
    from .weights import variable_specs
    specs = variable_specs(hp)
    cand, bn_have = {}, {}
    inside = []
    for name, e in entries.items():
        if not name.startswith(GRAPH_PREFIX):
            continue
        rel = name[len(GRAPH_PREFIX):]
        if rel.endswith('/Adam') or rel.endswith('/Adam_1') or rel == 'global_step':
            continue
        inside.append(name)
        can = _canonical(rel)
        m = _BN_RE.match(can)
        if m:
            bn_have.setdefault(m.group(1), {}).setdefault(int(m.group(2) or 0), {})[m.group(3)] = name
        else:
            cand.setdefault(can, []).append(name)
    mapping, problems = {}, []
    bn_want = {}
    for name, shape, _ in specs:
        can = _canonical(name)
        m = _BN_RE.match(can)
        if m:
            bn_want.setdefault(m.group(1), {}).setdefault(int(m.group(2) or 0), []).append((name, m.group(3), shape))
            continue
        names = cand.get(can, [])
        fits = [n for n in names if tuple(entries[n]['shape']) == tuple(shape)]
        if len(fits) == 1:
            mapping[name] = fits[0]
        elif not names:
            problems.append('missing %s %s' % (name, tuple(shape)))
        elif not fits:
            problems.append('shape %s: got %s want %s' % (name, ', '.join(str(tuple(entries[n]['shape'])) for n in names),
                                                          tuple(shape)))
        else:
            problems.append('ambiguous %s: %s' % (name, ', '.join(fits)))
    for scope, want in bn_want.items():
        have = bn_have.get(scope, {})
        have_order = sorted(have)
        if len(have_order) > len(want):
            # surplus batch-norm scopes: the k-th expected would silently take the k-th present one (all conv-bank BNs
            # have one shape), i.e. a checkpoint with more layers than hp would load into a different graph
            problems.append('scope %r holds %d batch_normalization scopes, the graph built from hp has %d' % (
                scope, len(have_order), len(want)))
        for k, idx in enumerate(sorted(want)):
            src = have[have_order[k]] if k < len(have_order) else {}
            for name, part, shape in want[idx]:
                n = src.get(part)
                if n is None:
                    problems.append('missing %s %s' % (name, tuple(shape)))
                elif tuple(entries[n]['shape']) != tuple(shape):
                    problems.append('shape %s: got %s (%s) want %s' % (name, tuple(entries[n]['shape']), n, tuple(shape)))
                else:
                    mapping[name] = n
    used = set(mapping.values())
    leftovers = [n for n in inside if n not in used]
    return mapping, problems, leftovers


# an unclaimed variable whose name has the form of a layer of the graph means the checkpoint was trained with MORE
# layers than hp describes (highway_5 with num_highway_layers=4, conv_bank_17, an extra prenet_dense_N or GRU cell):
# loading the rest would compute a different graph than the checkpoint's, so these raise by default
_LAYER_RE = re.compile(r'(^|/)(highway_\d+|prenet_dense_\d+|conv_bank_\d+|proj_\d+|cell_\d+|batch_normalization(_\d+)?|'
                       r'gru_cell|bidirectional_rnn|memory_layer|bahdanau_attention)/')


def load_tacotron_weights(prefix, hp=None, verify_crc=True, verbose=False, strict=False, report=None,
                          allow_extra_layers=False):
    """The inference-graph variables of a checkpoint the reference saved (model/tacotron.py:169; restored by
    synthesizer.py:32-34), as the {name relative to model/inference/: float32 array} dict the engine takes. Variables
    are matched by pattern and shape (match_variables). Missing or mis-shaped variables raise, with every problem and
    every unclaimed variable listed. Unclaimed variables that look like layers of the graph (highway_N, conv_bank_N,
    prenet_dense_N, batch_normalization_N, cell_N ...: a checkpoint with more layers than hp) raise too, unless
    allow_extra_layers; other unclaimed names (really foreign variables) are reported (`verbose` prints the whole
    name mapping; `report`, a dict, receives 'mapping', 'leftovers', 'renamed') and raise only under `strict`."""
    entries, _ = read_index(prefix)
    mapping, problems, leftovers = match_variables(entries, hp)
    if not allow_extra_layers:
        extra = [n for n in leftovers if _LAYER_RE.search(n[len(GRAPH_PREFIX):])]
        if extra:
            problems.append('the checkpoint has layers the graph built from hp does not (more layers than hp?): ' +
                            ', '.join(extra))
    renamed = {k: v for k, v in mapping.items() if v != GRAPH_PREFIX + k}
    if report is not None:
        report.update(mapping=dict(mapping), leftovers=list(leftovers), renamed=renamed)
    if verbose:
        for k in sorted(mapping):
            print('  %s <- %s%s' % (k, mapping[k], '' if k not in renamed else '   (renamed)'))
    left_txt = ['%s %s' % (n, tuple(entries[n]['shape'])) for n in leftovers]
    if problems:
        raise CheckpointError('checkpoint %s does not fit the inference graph:\n  %s%s' % (
            prefix, '\n  '.join(problems),
            ('\nvariables under %s that nothing claimed:\n  %s' % (GRAPH_PREFIX, '\n  '.join(left_txt))) if left_txt else ''))
    if left_txt:
        msg = 'variables under %s that the inference graph does not have:\n  %s' % (GRAPH_PREFIX, '\n  '.join(left_txt))
        if strict:
            raise CheckpointError(msg)
        print('load_tacotron_weights: ' + msg)
    raw = read_checkpoint(prefix, names=set(mapping.values()), verify_crc=verify_crc)
    return {k: np.asarray(raw[v], np.float32) for k, v in mapping.items()}


def save_tacotron_weights(prefix, w, spelling='tf1.15'):
    """Weights dict -> bundle under the reference's variable names (readable by tf.train.Saver). `spelling` selects
    among the name variants of SURVEY.md Appendix C that the loader accepts: 'tf1.15' (the names of weights.py),
    'residual_wrapper' (`cell_{1,2}/residual_wrapper/gru_cell/...`), 'global_bn' (batch-norm scopes numbered across
    the whole graph in creation order instead of per enclosing scope)."""
    from .weights import variable_specs  # noqa: F401  (creation order of the dict is the graph's)
    out, n_bn, seen = {}, 0, {}
    for k, v in w.items():
        name = k
        if spelling == 'residual_wrapper':
            name = re.sub(r'(multi_rnn_cell/cell_[12]/)(gru_cell/)', r'\1residual_wrapper/\2', name)
        elif spelling == 'global_bn':
            m = _BN_RE.match(name)
            if m:
                scope = m.group(1) + 'batch_normalization' + ('_' + m.group(2) if m.group(2) else '')
                if scope not in seen:
                    seen[scope] = n_bn
                    n_bn += 1
                name = '%sbatch_normalization%s/%s' % (m.group(1), '_%d' % seen[scope] if seen[scope] else '', m.group(3))
        elif spelling != 'tf1.15':
            raise ValueError('unknown spelling %r' % spelling)
        out[GRAPH_PREFIX + name] = np.asarray(v, np.float32)
    write_checkpoint(prefix, out)
