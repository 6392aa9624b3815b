"""Host-side mirror of pywfa's interface for the wavefront-alignment hot path.

``WavefrontAligner`` keeps the constructor kwargs, methods, properties, exceptions and result
objects of pywfa's Cython class (/root/reference/pywfa/align.pyx:306-883) and adds batch forms
(``wavefront_align_batch`` / ``align_batch``).  Every alignment — single pair or batch — runs on the
GPU through the C ABI of ``libwfa_hip.so`` (include/wfa_hip.h); there is no CPU path in this package
and constructing an aligner fails loudly when the library or a HIP device is missing.

Deliberate deviations from the reference (DESIGN.md §"Deviations"):
  * invalid penalties / ends-free sizes raise ``ValueError`` where WFA2-lib calls ``exit(1)``
    (wavefront_penalties.c:101-112, wavefront_align.c:95-101);
  * ``memory_mode="biwfa"``: ``scope="score"`` runs the ordinary score-only kernels (the reference returns the same
    scores there as in its other memory modes), ``scope="full"`` runs the breakpoint recursion on the device
    (csrc/wfa_biwfa.hpp); with a heuristic both directions of every breakpoint search cut their
    wavefronts off as the reference's do (wavefront_bialigner.c:53,161-166; both scopes); BiWFA with free ends raises ``NotImplementedError`` (the reference itself
    ``exit(1)``s, wavefront_align.c:60-75);
  * property setters re-derive the whole native configuration (the reference pokes single C fields
    and leaves derived state stale).
"""
import sys

import numpy as np

from . import _native
from . import datagen

_NO_OPS = np.zeros(0, np.uint8)   # (the op string of a call without a backtrace: one read-only array for every such call)
_NO_OPS.flags.writeable = False

__all__ = ["WavefrontAligner", "AlignmentResult", "BatchResults", "clip_cigartuples", "cigartuples_to_str",
           "elide_mismatches_from_cigar"]

# CIGAR tuple codes (align.pyx:11-14, README.rst:61-86): M I D N S H P = X B
_OP_CODE = {ord("M"): 0, ord("I"): 1, ord("D"): 2, ord("N"): 3, ord("S"): 4, ord("H"): 5,
            ord("P"): 6, ord("="): 7, ord("X"): 8, ord("B"): 9}
_OP_CHARS = "MIDNSHP=XB"
_INT_MAX = 2147483647


class AlignmentResult:
    """Result object of ``WavefrontAligner.__call__`` (align.pyx:17-180)."""

    def __init__(self, pl, tl, ps, pe, ts, te, ct, s, p, t, status):
        self.pattern_length = pl
        self.text_length = tl
        self.pattern_start = ps
        self.pattern_end = pe
        self.text_start = ts
        self.text_end = te
        self.cigartuples = ct
        self.score = s
        self.pattern = p
        self.text = t
        self.status = status

    def __repr__(self):
        keys = ("score", "pattern_start", "pattern_end", "text_start", "text_end", "cigartuples",
                "pattern", "text")
        return "".join(f"    {k}: {self.__dict__[k]}\n" for k in keys)

    def __str__(self):
        score = "Score: %d" % self.score
        if self.pattern and self.cigartuples:
            t = self.aligned_text
            p = self.aligned_pattern
            if len(t) > 30:
                t = t[:30] + "..."
                p = p[:30] + "..."
            c = self.cigarstring[:30]
            return "\n".join([p, t, c, score, "Length: %d" % len(t)])
        return score

    def __eq__(self, other):
        return isinstance(other, AlignmentResult) and self.__dict__ == other.__dict__

    @staticmethod
    def _aligned(sequence, tuples, begin, end, gap_type):
        """``aligned_pattern`` / ``aligned_text``.  The reference reads every
        ``(code, run)`` tuple as ``(width, kind)`` and inserts a gap only where ``kind`` equals the gap letter; a run
        length is never a letter, so for tuples the aligner produced no gap is inserted, the pieces tile the window
        ``sequence[begin:end]`` and the result is that window.  Hand-made tuples whose second field IS the gap letter
        still get their dashes."""
        window = sequence[begin:end]
        if all(kind != gap_type for _, kind in tuples):
            return window
        pieces, at = [], 0
        for width, kind in tuples:
            if kind == gap_type:
                pieces.append("-" * width)
                continue
            pieces.append(window[at:at + width])
            at += width
        return "".join(pieces) + window[at:end - begin]

    @property
    def aligned_pattern(self):
        if self.pattern:
            return self._aligned(self.pattern, self.cigartuples, self.pattern_start, self.pattern_end, "D")
        return None

    @property
    def aligned_text(self):
        if self.text:
            return self._aligned(self.text, self.cigartuples, self.text_start, self.text_end, "I")
        return None

    @property
    def cigarstring(self):
        return cigartuples_to_str(self.cigartuples)

    # rows of the three-line view per cigartuple code: (takes pattern bases, takes text bases, glyph of the middle row);
    # M / = are joined by '|', X by '*', gaps and clips by blanks, anything else is refused
    _PRETTY_ROWS = {0: (True, True, "|"), 7: (True, True, "|"), 8: (True, True, "*"),
                    1: (False, True, " "), 4: (False, True, " "), 5: (False, True, " "),
                    2: (True, False, " ")}

    @property
    def pretty(self):
        """The CIGAR, the CIGAR without its match runs, and the PATTERN / marks / TEXT rows."""
        rows = {"p": "      PATTERN    ", "g": "                 ", "t": "      TEXT       "}
        used_p = used_t = 0
        for code, run in self.cigartuples:
            if code not in self._PRETTY_ROWS:
                raise ValueError(f"Cigar operation not available for pretty print - {code}")
            on_p, on_t, glyph = self._PRETTY_ROWS[code]
            rows["p"] += self.pattern[used_p:used_p + run] if on_p else "-" * run
            rows["t"] += self.text[used_t:used_t + run] if on_t else "-" * run
            rows["g"] += glyph * run
            used_p += run if on_p else 0
            used_t += run if on_t else 0
        # (the compact form drops the match runs only: the reference's second filter compares an int with a list)
        compact = cigartuples_to_str([ct for ct in self.cigartuples if ct[0] != 0])
        head = f"{self.cigarstring}      ALIGNMENT\n{compact}      ALIGNMENT.COMPACT\n"
        return head + rows["p"] + "\n" + rows["g"] + "\n" + rows["t"] + "\n"


def _flank_scan(ct, threshold_left, threshold_right, text_len, pattern_len):
    """Shared scan of clip_cigartuples / locations: skip ops at both flanks until an M run of at
    least the threshold (align.pyx:199-234, :797-831).  Returns (i, j, ps, pe, ts, te)."""
    ts = ps = 0
    i = 0
    for i in range(len(ct)):
        op, n = ct[i]
        if op == 0:
            if n >= threshold_left:
                break
            ts += n; ps += n
        elif op == 2:
            ps += n
        elif op == 8:
            ts += n; ps += n
        elif op == 1:
            ts += n
    te, pe = text_len, pattern_len
    j = len(ct) - 1
    for j in range(len(ct) - 1, -1, -1):
        op, n = ct[j]
        if op == 0:
            if n >= threshold_right:
                break
            te -= n; pe -= n
        elif op == 2:
            pe -= n
        elif op == 8:
            pe -= n; te -= n
        elif op == 1:
            te -= n
    return i, j, ps, pe, ts, te


def clip_cigartuples(align_result, min_aligned_bases_left=5, min_aligned_bases_right=5):
    """Trim flanking blocks with fewer aligned bases than the thresholds into soft-clips
    (align.pyx:183-250).  Mutates and returns ``align_result``; the score is not adjusted."""
    ct = align_result.cigartuples
    if not ct:
        return align_result
    i, j, ps, pe, ts, te = _flank_scan(ct, int(min_aligned_bases_left), int(min_aligned_bases_right),
                                       align_result.text_length, align_result.pattern_length)
    modified = []
    if align_result.text_start + ts > 0:
        modified.append((4, ts))
    modified += ct[i:j + 1]
    if align_result.text_length - te > 0:
        modified.append((4, align_result.text_length - te))
    align_result.cigartuples = modified
    align_result.text_start, align_result.text_end = ts, te
    align_result.pattern_start, align_result.pattern_end = ps, pe
    return align_result


def elide_mismatches_from_cigar(cigartuples):
    """Merge adjacent M (0) and X (8) runs into single M runs (align.pyx:253-277)."""
    if not cigartuples:
        return []
    out = []
    block = 0
    for op, n in cigartuples:
        if op == 0 or op == 8:
            block += n
        else:
            if block:
                out.append((0, block))
                block = 0
            out.append((op, n))
    if block:
        out.append((0, block))
    return out


def cigartuples_to_str(cigartuples):
    """``[(op, n), ...]`` -> ``"{n}{op}..."`` (align.pyx:280-295)."""
    if not cigartuples:
        return ""
    return "".join(f"{int(n)}{_OP_CHARS[op]}" for op, n in cigartuples)


def _rle(ops):
    """Run-length encode a uint8 array of op chars -> (chars, lengths)."""
    if ops.size == 0:
        return ops, np.zeros(0, np.int64)
    change = np.flatnonzero(ops[1:] != ops[:-1]) + 1
    starts = np.concatenate(([0], change))
    lengths = np.diff(np.concatenate((starts, [ops.size])))
    return ops[starts], lengths


def _ops_to_tuples(ops):
    ch, ln = _rle(ops)
    return [(_OP_CODE[int(c)], int(n)) for c, n in zip(ch, ln)]


def _ops_to_string(ops):
    ch, ln = _rle(ops)
    return "".join(f"{int(n)}{chr(int(c))}" for c, n in zip(ch, ln))


class _RunSequence:
    """Read-only sequence over the per-pair CIGAR runs of a batch (run-length encoded on the GPU: ``run_off`` int64[n+1],
    ``run_code`` uint8 = pywfa's cigartuple codes, ``run_len`` int32).  Items are built on access — a Python str per pair
    costs ~1 us, the alignment itself ~0.001 us — ``kind``: "str" = CIGAR string, "ops" = uint8 array of op characters."""

    _CHARS = np.frombuffer(_OP_CHARS.encode(), dtype=np.uint8)   # cigartuple code -> op character

    def __init__(self, run_off, run_code, run_len, kind):
        self._off, self._code, self._len, self._kind = run_off, run_code, run_len, kind

    def __len__(self):
        return len(self._off) - 1

    def _item(self, i):
        a, b = int(self._off[i]), int(self._off[i + 1])
        if self._kind == "str":
            return "".join(f"{n}{_OP_CHARS[c]}" for c, n in zip(self._code[a:b].tolist(), self._len[a:b].tolist()))
        return np.repeat(self._CHARS[self._code[a:b]], self._len[a:b])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._item(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self._item(i)

    def __iter__(self):
        return (self._item(i) for i in range(len(self)))


class _OpsSequence:
    """Read-only sequence over the per-pair op strings of a small batch (``ops`` uint8, ``begin`` / ``length`` per pair);
    ``kind``: "str" = CIGAR string (run-length encoded when read), "ops" = uint8 array of op characters."""

    def __init__(self, ops, begin, length, kind):
        self._ops, self._beg, self._len, self._kind = ops, begin, length, kind

    def __len__(self):
        return len(self._beg)

    def _item(self, i):
        o = self._ops[self._beg[i]:self._beg[i] + self._len[i]]
        return _ops_to_string(o) if self._kind == "str" else o

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._item(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self._item(i)

    def __iter__(self):
        return (self._item(i) for i in range(len(self)))


class BatchResults:
    """Results of a whole batch with the Python-side surface pre-computed on the device: ``score``,
    ``status``, run-length encoded CIGARs (``run_off``, ``run_code``, ``run_len``) and ``locations``
    (n x 4: pattern_start, pattern_end, text_start, text_end).  ``res[i]`` builds the AlignmentResult
    that ``WavefrontAligner.__call__`` would return for pair i."""

    def __init__(self, batch, score, status, run_off, run_code, run_len, locations):
        self._batch = batch
        self.score, self.status = score, status
        self.run_off, self.run_code, self.run_len = run_off, run_code, run_len
        self.locations = locations

    def __len__(self):
        return len(self.score)

    def cigartuples(self, i):
        a, b = int(self.run_off[i]), int(self.run_off[i + 1])
        return [(int(c), int(n)) for c, n in zip(self.run_code[a:b], self.run_len[a:b])]

    def cigarstring(self, i):
        return cigartuples_to_str(self.cigartuples(i))

    def __getitem__(self, i):
        p, t = datagen.pair_strings(self._batch, i)
        ps, pe, ts, te = (int(x) for x in self.locations[i])
        return AlignmentResult(len(p), len(t), ps, pe, ts, te, self.cigartuples(i), int(self.score[i]), p, t,
                               int(self.status[i]))


class WavefrontAligner:
    """Drop-in for ``pywfa.WavefrontAligner`` on the GPU. If a pattern is supplied it is cached.

    **Cost of a call.**  Every alignment, a single pair included, runs on the GPU (there is no CPU path).  One
    ``wavefront_align(text)`` of 150 bp reads costs about 14 us at the C ABI (``wfa_hip_align_pair``: host packs the pair into a
    pinned block, one wave aligns it, the host polls a completion flag), 19 us with the op string, a few us more through this
    class — against 1-2 us for pywfa on a host core.  A loop of single calls is exact but 10x slower than the reference's; the
    additive batch forms (``wavefront_align_batch``, ``align_batch``, ``align_batch_results``) are what this class is for:
    from about 32 pairs per call the GPU is ahead of one host core, at 10 M pairs per call by 300x (3 G alignments/s resident,
    0.3 G/s from host memory).
    """

    def __init__(self, pattern=None, distance="affine", memory_mode="high", match=0, mismatch=4,
                 gap_opening=6, gap_extension=2, gap_opening2=24, gap_extension2=1, scope="full",
                 span="ends-free", pattern_begin_free=0, pattern_end_free=0, text_begin_free=0,
                 text_end_free=0, heuristic=None, min_wavefront_length=10,
                 max_distance_threshold=50, steps_between_cutoffs=1, xdrop=20, wildcard=None,
                 max_steps=0, device=0, devices=None):
        self.pattern_len = 0
        self.text_len = 0
        self.alignment_score = 0
        self._pattern = None
        self._bpattern = None
        self._text = None
        if pattern:
            self._set_pattern(pattern)
        self._wildcard = None
        self._bwildcard = -1
        self.wildcard = wildcard
        cfg = _native.default_config()
        if distance not in _native.DIST:
            raise NotImplementedError(f"{distance} distance not implemented")
        cfg.distance = _native.DIST[distance]
        cfg.match, cfg.mismatch = int(match), int(mismatch)
        cfg.gap_opening, cfg.gap_extension = int(gap_opening), int(gap_extension)
        cfg.gap_opening2, cfg.gap_extension2 = int(gap_opening2), int(gap_extension2)
        if scope not in _native.SCOPE:
            raise ValueError(f"{scope} scope not understood")
        cfg.scope = _native.SCOPE[scope]
        if memory_mode not in _native.MEM:
            raise ValueError("memory_mode must be one of 'high', 'medium', 'low', 'biwfa'")
        cfg.memory_mode = _native.MEM[memory_mode]
        cfg.pattern_begin_free, cfg.pattern_end_free = int(pattern_begin_free), int(pattern_end_free)
        cfg.text_begin_free, cfg.text_end_free = int(text_begin_free), int(text_end_free)
        if span not in _native.SPAN:
            raise NotImplementedError(f"{span} span not implemented")
        cfg.span = _native.SPAN[span]
        if heuristic not in _native.HEUR:
            raise NotImplementedError(f"{heuristic} heuristic not implemented")
        cfg.heuristic = _native.HEUR[heuristic]
        cfg.min_wavefront_length = int(min_wavefront_length)
        cfg.max_distance_threshold = int(max_distance_threshold)
        cfg.steps_between_cutoffs = int(steps_between_cutoffs)
        cfg.xdrop = int(xdrop)
        cfg.max_steps = int(max_steps) if int(max_steps) > 0 else 0
        cfg.wildcard = self._bwildcard
        self._cfg = cfg
        self._native = _native.Aligner(cfg, device)  # raises if no library / no GPU / bad config
        self._host_scratch = {}   # buffers of the compiled host's batch marshalling (pywfa_amd/host/_host.pyx), kept between calls
        # devices=[...] (additive): batches given to wavefront_align_batch / align_batch are sharded over these GPUs of the
        # node (contiguous shards balanced by bases, one host thread per device, no collective); single pairs and resident
        # batches stay on `device`
        self._multi = _native.MultiAligner(cfg, devices) if devices is not None and len(devices) > 1 else None
        # last single-pair result (the reference keeps it inside the C aligner object)
        self._status = -1
        self._score = -2147483648
        self._ops = _NO_OPS

    # ------------------------------------------------------------------ helpers
    def _set_pattern(self, pattern):
        self._pattern = pattern.upper()
        self._bpattern = self._pattern.encode("ascii")
        self.pattern_len = len(self._bpattern)

    def _push(self):
        self._cfg.wildcard = self._bwildcard
        self._native.set_config(self._cfg)
        if self._multi is not None:
            self._multi.set_config(self._cfg)

    # ------------------------------------------------------------------ single pair
    def wavefront_align(self, text, pattern=None):
        """Align one pair; returns the score (align.pyx:421-443)."""
        if pattern is not None:
            self._set_pattern(pattern)
        t = text.upper().encode("ascii")
        self._text = text
        self.text_len = len(t)
        if self._bpattern is None:
            raise AttributeError("pattern has not been set")
        if self._cfg.wildcard != self._bwildcard:
            self._push()
        # one pair per call: wfa_hip_align_pair (no arrays on the way; about 14 us per 150 bp call at the C ABI, 19 us with the op
        # string — the reference takes 1-2 us on a host core: loops of single calls work and are exact, throughput needs
        # wavefront_align_batch)
        full = self._cfg.scope == 1
        score, status, ops = self._native.align_pair(self._bpattern, t, full)
        self._score = score
        self._status = status
        self._ops = np.frombuffer(ops, np.uint8) if ops else _NO_OPS
        self.alignment_score = self._score
        return self._score

    def __call__(self, text, pattern=None, clip_cigar=False, min_aligned_bases_left=1,
                 min_aligned_bases_right=1, elide_mismatches=False, supress_sequences=False):
        """Align ``text`` to ``pattern`` and return an AlignmentResult (align.pyx:835-879)."""
        if pattern is None:
            p = self._pattern
            if not p:
                raise ValueError("pattern is None")
            lp = len(self._pattern)
            score = self.wavefront_align(text)
        else:
            lp = len(pattern)
            p = pattern
            score = self.wavefront_align(text, pattern)
        ct = self.cigartuples
        locs = self.locations
        status = self.status
        if supress_sequences:
            res = AlignmentResult(lp, len(text), locs[0], locs[1], locs[2], locs[3], ct, score, "", "", status)
        else:
            res = AlignmentResult(lp, len(text), locs[0], locs[1], locs[2], locs[3], ct, score, p, text, status)
        # as committed in the reference the test is inverted: clip / elide only act when the scope is
        # NOT "full", i.e. on an empty CIGAR (align.pyx:874-878)
        if not self.scope == "full":
            if clip_cigar:
                res = clip_cigartuples(res, min_aligned_bases_left, min_aligned_bases_right)
            if elide_mismatches:
                res.cigartuples = elide_mismatches_from_cigar(res.cigartuples)
        return res

    # ------------------------------------------------------------------ batches (additive API)
    def wavefront_align_batch(self, texts, patterns=None):
        """Align many pairs on the GPU. ``patterns`` None = the cached pattern for every text.

        Returns dict(score=int32[n], status=int32[n], cigarstrings=sequence of str, cigar_ops=sequence of uint8 arrays
        (scope full; built from the GPU's run-length encoding when an item is read))."""
        texts = texts if type(texts) is list else list(texts)
        if patterns is None:
            if self._bpattern is None:
                raise ValueError("pattern is None")
            patterns = self._bpattern  # stored once in the batch: every pair points to it
        else:
            patterns = patterns if type(patterns) is list else list(patterns)
        if self._cfg.wildcard != self._bwildcard:
            self._push()
        # the compiled host reads the objects' buffers in place and upper-cases on OpenMP threads (pywfa_amd/host/_host.pyx); objects
        # it does not take (non-ASCII text, other types) and builds without the extension go through datagen.from_strings, which
        # raises what the reference raises (align.pyx:432,435)
        host = _native.compiled_host()
        # (its blob and offset arrays are kept between calls: align_batch below returns only after the upload, and keeps nothing of the batch)
        batch = host.from_strings(patterns, texts, self._host_scratch) if host is not None else None
        if batch is None:
            batch = datagen.from_strings(patterns, texts, upper=True)
        return self.align_batch(batch)

    def align_batch(self, batch):
        """Align a prepared batch dict (see ``pywfa_amd.datagen``): ASCII blob + offsets + lengths."""
        full = self._cfg.scope == 1
        if full and self._multi is None and len(batch["p_len"]) <= 1024:
            # a small batch: the single-call form of the library (one launch, results polled in a pinned block); the CIGAR
            # strings are run-length encoded when they are read
            score, status, (ops, cbeg, clen) = self._native.align_batch(batch, True)
            return {"score": score, "status": status, "cigar_ops": _OpsSequence(ops, cbeg, clen, "ops"),
                    "cigarstrings": _OpsSequence(ops, cbeg, clen, "str")}
        if full and self._multi is None:
            # one device: the op strings stay on the GPU, their run-length encoding comes back (csrc/wfa_rle.hpp); the Python
            # strings / op arrays are built when they are read
            rb = self._native.batch(batch)
            try:
                rb.run()
                rb.sync()
                score, status, _ = rb.results(False)
                off, code, rlen, _locs = rb.rle()
            finally:
                rb.close()
            return {"score": score, "status": status, "cigar_ops": _RunSequence(off, code, rlen, "ops"),
                    "cigarstrings": _RunSequence(off, code, rlen, "str")}
        score, status, cig = (self._multi or self._native).align_batch(batch, full)
        out = {"score": score, "status": status}
        if full:
            ops, cbeg, clen = cig
            out["cigar_ops"] = [ops[cbeg[i]:cbeg[i] + clen[i]] for i in range(len(score))]
            out["cigarstrings"] = [_ops_to_string(o) for o in out["cigar_ops"]]
        return out

    def align_batch_results(self, batch, patterns_texts=None):
        """Align a batch (scope must be "full") and return a ``BatchResults``: scores, statuses, and the
        cigartuples / locations of every pair computed on the GPU (what ``__call__`` derives per pair)."""
        if self._cfg.scope != 1:
            raise ValueError("align_batch_results needs scope='full'")
        rb = self._native.batch(batch)
        try:
            rb.run()
            rb.sync()
            score, status, _ = rb.results(False)
            off, code, rlen, locs = rb.rle()
        finally:
            rb.close()
        return BatchResults(batch, score, status, off, code, rlen, locs)

    def resident_batch(self, batch):
        """Upload + 2-bit pack a batch into HBM once; ``.run()`` it many times (bench.py)."""
        return self._native.batch(batch)

    # ------------------------------------------------------------------ results
    @property
    def status(self):
        return self._status

    @property
    def score(self):
        return self._score

    @property
    def cigarstring(self):
        return _ops_to_string(self._ops)

    @property
    def cigartuples(self):
        return _ops_to_tuples(self._ops)

    @property
    def locations(self):
        """(pattern_start, pattern_end, text_start, text_end) (align.pyx:788-833)."""
        if self.scope == "score":
            return [0, 0, 0, 0]
        ct = self.cigartuples
        if not ct or self.text_len == 0 or self.pattern_len == 0:
            return [0, 0, 0, 0]
        _, _, ps, pe, ts, te = _flank_scan(ct, 1, 1, self.text_len, self.pattern_len)
        return ps, pe, ts, te

    def cigar_print_pretty(self, file_name=None):
        """ALIGNMENT / ETRACE / CIGAR + three alignment rows: the text of cigar_print_pretty
        (WFA2_lib/alignment/cigar.c:778-863, called by align.pyx:445-459) from the C ABI's wfa_hip_cigar_sprint_pretty."""
        out = _native.cigar_sprint_pretty(self._ops, self._bpattern, self._text.encode("ascii"))
        if file_name:
            with open(file_name, "w") as f:
                f.write(out)
        else:
            sys.stdout.write(out)

    # ------------------------------------------------------------------ configuration properties
    def _int_prop(name):  # noqa: N805
        def getter(self):
            return getattr(self._cfg, name)

        def setter(self, value):
            old = getattr(self._cfg, name)
            setattr(self._cfg, name, int(value))
            try:
                self._push()
            except Exception:
                setattr(self._cfg, name, old)
                raise
        return property(getter, setter)

    pattern_begin_free = _int_prop("pattern_begin_free")
    pattern_end_free = _int_prop("pattern_end_free")
    text_begin_free = _int_prop("text_begin_free")
    text_end_free = _int_prop("text_end_free")
    min_wavefront_length = _int_prop("min_wavefront_length")
    max_distance_threshold = _int_prop("max_distance_threshold")
    steps_between_cutoffs = _int_prop("steps_between_cutoffs")
    xdrop = _int_prop("xdrop")
    mismatch_penalty = _int_prop("mismatch")
    gap_opening_penalty = _int_prop("gap_opening")
    gap_extension_penalty = _int_prop("gap_extension")
    gap_opening2_penalty = _int_prop("gap_opening2")
    gap_extension2_penalty = _int_prop("gap_extension2")
    match_score = _int_prop("match")
    del _int_prop

    def _enum_prop(name, table, exc, msg, canon=None):  # noqa: N805
        inv = {v: k for k, v in (canon or table).items()}

        def getter(self):
            return inv[getattr(self._cfg, name)]

        def setter(self, value):
            if value not in table:
                raise exc(msg.format(value))
            old = getattr(self._cfg, name)
            setattr(self._cfg, name, table[value])
            try:
                self._push()
            except Exception:
                setattr(self._cfg, name, old)
                raise
        return property(getter, setter)

    scope = _enum_prop("scope", _native.SCOPE, ValueError, "{} scope not understood")
    span = _enum_prop("span", _native.SPAN, NotImplementedError, "{} span not implemented")
    heuristic = _enum_prop("heuristic", _native.HEUR, NotImplementedError, "{} heuristic not implemented")
    distance = _enum_prop("distance", _native.DIST, NotImplementedError, "{} distance not implemented")
    # the reference's setter accepts "med" where the constructor wants "medium" (align.pyx:547-549)
    memory_mode = _enum_prop("memory_mode", dict(_native.MEM, med=1), NotImplementedError,
                             "{} memory_mode not implemented", canon=_native.MEM)
    del _enum_prop

    @property
    def wildcard(self):
        return self._wildcard

    @wildcard.setter
    def wildcard(self, wildcard):
        if wildcard is not None:
            if not isinstance(wildcard, str):
                raise TypeError(f"expected wildcard to be a string, but it is {type(wildcard)}")
            if len(wildcard) > 1:
                raise ValueError(f"wildcard must have length 1, but has length {len(wildcard)}")
            self._wildcard = wildcard
            self._bwildcard = wildcard.upper().encode("ascii")[0]
        else:
            self._wildcard = None
            self._bwildcard = -1

    @property
    def max_steps(self):
        return self._cfg.max_steps if self._cfg.max_steps > 0 else _INT_MAX

    @max_steps.setter
    def max_steps(self, steps):
        steps = int(steps)
        self._cfg.max_steps = steps if 0 < steps < _INT_MAX else 0
        self._push()

    def close(self):
        self._native.close()
        if self._multi is not None:
            self._multi.close()
