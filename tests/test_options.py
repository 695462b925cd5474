"""Options tree: defaults, validation and the protobuf wire format of the
reference (nufft_options_test.py:22-48, proto/nufft_options.proto:19-32)."""
import pytest

import tensorflow_nufft as tfft
from tensorflow_nufft import _proto


def _reference_message_classes():
  """Builds the reference's messages with the protobuf runtime (no protoc)."""
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto()
  fd.name = 'nufft_options_test.proto'
  fd.package = 'tensorflow.nufft.testcopy'
  fd.syntax = 'proto3'
  e = fd.enum_type.add(); e.name = 'FftwPlanningRigor'
  for i, n in enumerate(['AUTO', 'ESTIMATE', 'MEASURE', 'PATIENT', 'EXHAUSTIVE']):
    v = e.value.add(); v.name = n; v.number = i
  e = fd.enum_type.add(); e.name = 'PointsRange'
  for i, n in enumerate(['STRICT', 'EXTENDED', 'INFINITE']):
    v = e.value.add(); v.name = n; v.number = i
  m = fd.message_type.add(); m.name = 'FftwOptions'
  f = m.field.add(); f.name = 'planning_rigor'; f.number = 1; f.type = 14; f.label = 1
  f.type_name = '.tensorflow.nufft.testcopy.FftwPlanningRigor'
  m = fd.message_type.add(); m.name = 'DebuggingOptions'
  f = m.field.add(); f.name = 'check_points_range'; f.number = 1; f.type = 8; f.label = 1
  m = fd.message_type.add(); m.name = 'Options'
  f = m.field.add(); f.name = 'debugging'; f.number = 1; f.type = 11; f.label = 1
  f.type_name = '.tensorflow.nufft.testcopy.DebuggingOptions'
  f = m.field.add(); f.name = 'fftw'; f.number = 2; f.type = 11; f.label = 1
  f.type_name = '.tensorflow.nufft.testcopy.FftwOptions'
  f = m.field.add(); f.name = 'max_batch_size'; f.number = 3; f.type = 5; f.label = 1
  f = m.field.add(); f.name = 'points_range'; f.number = 4; f.type = 14; f.label = 1
  f.type_name = '.tensorflow.nufft.testcopy.PointsRange'
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.nufft.testcopy.Options'))


def test_defaults():
  o = tfft.Options()
  assert o.points_range == tfft.PointsRange.EXTENDED
  assert o.debugging.check_points_range is False
  assert o.max_batch_size is None
  assert o.fftw.planning_rigor == tfft.FftwPlanningRigor.AUTO


def test_validation_on_assignment():
  o = tfft.Options()
  with pytest.raises(Exception):
    o.points_range = 7
  with pytest.raises(Exception):
    o.max_batch_size = 'many'
  o.points_range = 'INFINITE' if False else tfft.PointsRange.INFINITE
  o.max_batch_size = 4


def test_proto_round_trip():
  o = tfft.Options()
  o.debugging.check_points_range = True
  o.fftw.planning_rigor = tfft.FftwPlanningRigor.PATIENT
  o.max_batch_size = 300
  o.points_range = tfft.PointsRange.STRICT
  data = o.to_proto().SerializeToString()
  back = tfft.Options.from_proto(_proto.OptionsProto().ParseFromString(data))
  assert back == o
  assert tfft.Options.from_proto(tfft.Options().to_proto()) == tfft.Options()


def test_wire_format_matches_protobuf_runtime():
  Msg = _reference_message_classes()
  for check, rigor, mbs, pr in [(False, 0, None, 1), (True, 3, 2, 2), (True, 0, 300, 0), (False, 4, 1 << 20, 1)]:
    o = tfft.Options()
    o.debugging.check_points_range = check
    o.fftw.planning_rigor = tfft.FftwPlanningRigor(rigor)
    o.max_batch_size = mbs
    o.points_range = tfft.PointsRange(pr)
    ours = o.to_proto().SerializeToString()
    m = Msg()
    m.debugging.check_points_range = check
    m.fftw.planning_rigor = rigor
    if mbs is not None:
      m.max_batch_size = mbs
    m.points_range = pr
    assert ours == m.SerializeToString(deterministic=True)
    # and the runtime's bytes parse back into the same options
    assert tfft.Options.from_proto(m.SerializeToString()) == o


# ---- the library's own decoder of the `options` attr (nufft_hip_options_from_proto), which
# replaces Options::ParseFromString in the reference kernel (nufft_kernels.cc:582-585) ----

def _c_decode(data):
  import ctypes
  from tensorflow_nufft import _lib
  o = _lib.OptionsStruct()
  o.kernel_width = 77   # must be overwritten with the default on success, kept on failure
  rc = _lib.lib().nufft_hip_options_from_proto(bytes(data), len(data), ctypes.byref(o))
  return rc, o


def _expect(o, check, rigor, mbs, pr):
  assert (o.check_points_range, o.fftw_planning_rigor, o.max_batch_size, o.points_range) == \
      (int(check), rigor, mbs, pr)
  # everything outside the proto keeps its nufft_hip_default_options value
  assert (o.spread_only, o.kernel_width, o.upsampling_factor, o.spread_method,
          o.max_subproblem_size, list(o.tile_dims), o.lds_accumulate, o.num_point_sets) == \
      (0, 0, 0.0, 0, 0, [0, 0, 0], 0, 0)


def test_c_decoder_matches_protobuf_runtime_on_every_field_combination():
  Msg = _reference_message_classes()
  n = 0
  for check in (None, False, True):            # None = sub-message absent
    for rigor in (None, 0, 1, 4):
      for mbs in (None, 0, 1, 127, 128, 300, 1 << 20, 2**31 - 1, -1, -5):
        for pr in (None, 0, 1, 2):
          m = Msg()
          if check is not None:
            m.debugging.check_points_range = check
            m.debugging.SetInParent()
          if rigor is not None:
            m.fftw.planning_rigor = rigor
            m.fftw.SetInParent()
          if mbs is not None:
            m.max_batch_size = mbs
          if pr is not None:
            m.points_range = pr
          data = m.SerializeToString()
          rc, o = _c_decode(data)
          assert rc == 0, data
          back = Msg.FromString(data)   # what the reference kernel sees after ParseFromString
          _expect(o, back.debugging.check_points_range, back.fftw.planning_rigor,
                  back.max_batch_size, back.points_range)
          n += 1
  assert n == 3 * 4 * 10 * 4


def test_c_decoder_proto3_defaults_and_python_wrapper_default():
  # empty attr (`options: string = ''`): every field at the proto3 zero => STRICT
  rc, o = _c_decode(b'')
  assert rc == 0
  _expect(o, False, 0, 0, 0)
  # the Python wrapper always sends EXTENDED explicitly (nufft_ops.py:118-123)
  rc, o = _c_decode(tfft.Options().to_proto().SerializeToString())
  assert rc == 0
  _expect(o, False, 0, 0, 1)


def test_c_decoder_unknown_fields_repeats_and_foreign_wire_types():
  Msg = _reference_message_classes()
  m = Msg(); m.max_batch_size = 9; m.points_range = 2; m.debugging.check_points_range = True
  base = m.SerializeToString()
  unknown = (b'\x28\x96\x01'               # field 5 varint 150
             b'\x32\x03abc'                # field 6 bytes
             b'\x39' + b'\x01' * 8 +       # field 7 fixed64
             b'\x45' + b'\x02' * 4 +       # field 8 fixed32
             b'\x4b\x50\x01\x4c'           # field 9 group { field 10 varint } end group
             b'\xf8\xff\xff\xff\x0f\x01')  # field 2^29-1 varint
  for data in (unknown + base, base + unknown, base[:2] + unknown + base[2:] if False else base + unknown + base):
    rc, o = _c_decode(data)
    assert rc == 0
    back = Msg.FromString(data)
    _expect(o, back.debugging.check_points_range, back.fftw.planning_rigor, back.max_batch_size,
            back.points_range)
    assert o.max_batch_size == 9 and o.points_range == 2 and o.check_points_range == 1
  # last value wins for scalars; sub-messages merge
  data = b'\x18\x05\x18\x07' + b'\x0a\x02\x08\x01' + b'\x0a\x00' + b'\x20\x02\x20\x01'
  rc, o = _c_decode(data)
  back = Msg.FromString(data)
  assert rc == 0 and back.max_batch_size == 7 and back.debugging.check_points_range
  _expect(o, True, 0, 7, 1)
  # a known field number with a foreign wire type is an unknown field, not an error
  data = b'\x1a\x02hi' + b'\x25' + b'\x00' * 4 + b'\x08\x01' + b'\x18\x03'
  rc, o = _c_decode(data)
  back = Msg.FromString(data)
  assert rc == 0 and back.max_batch_size == 3 and back.points_range == 0
  _expect(o, False, 0, 3, 0)
  # unknown fields inside the sub-messages, 64-bit varints truncate to int32 like the runtime
  data = b'\x0a\x06\x10\x05\x08\x01\x1a\x00' + b'\x18\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01'
  rc, o = _c_decode(data)
  back = Msg.FromString(data)
  assert rc == 0 and back.max_batch_size == -1
  _expect(o, True, 0, -1, 0)


def test_c_decoder_refuses_what_the_runtime_refuses():
  from google.protobuf.message import DecodeError
  Msg = _reference_message_classes()
  m = Msg(); m.max_batch_size = 300; m.points_range = 1; m.debugging.check_points_range = True
  m.fftw.planning_rigor = 3
  good = m.SerializeToString()
  bad = [good[:k] for k in range(1, len(good))]       # every truncation ...
  bad = [b for b in bad if not _parses(Msg, b)]       # ... that is not itself a valid message
  assert len(bad) >= 4
  bad += [b'\x18',                                    # key without value
          b'\x18' + b'\xff' * 10 + b'\x01',           # varint longer than 10 bytes
          b'\x0a\x05\x08\x01',                        # sub-message longer than the buffer
          b'\x0a\x02\x08',                            # truncated inside the sub-message
          b'\x00\x01',                                # field number 0
          b'\x1c',                                    # stray end-group
          b'\x4b\x50\x01',                            # group never closed
          b'\x4b\x50\x01\x54',                        # group closed by another field's end tag
          b'\x1e\x00', b'\x1f\x00',                   # wire types 6 and 7
          b'\x39\x01\x02']                            # fixed64 cut short
  for data in bad:
    with pytest.raises(DecodeError):
      Msg.FromString(data)
    rc, o = _c_decode(data)
    assert rc == 3, data          # NUFFT_HIP_INVALID_ARGUMENT
    assert o.kernel_width == 77   # *out untouched


def _parses(Msg, data):
  from google.protobuf.message import DecodeError
  try:
    Msg.FromString(data)
    return True
  except DecodeError:
    return False


def test_c_decoder_agrees_with_the_runtime_on_random_bytes():
  import numpy as np
  Msg = _reference_message_classes()
  rng = np.random.default_rng(7)
  # bytes biased towards this schema's tags so that a fair share parses
  alphabet = np.frombuffer(b'\x0a\x12\x18\x20\x08\x00\x01\x02\x04\x7f\x80\xff\x28\x32\x1a\x25\x0c\x21', np.uint8)
  ok = 0
  for _ in range(6000):
    n = int(rng.integers(0, 12))
    data = bytes(rng.choice(alphabet, n)) if rng.random() < .8 else bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if any((b & 7) == 3 for b in data):
      # a possible start-group tag: the Python (upb) runtime skips unknown groups more leniently than
      # the C++ runtime the reference links (which refuses field number 0 inside them, as this
      # decoder does); groups are covered by the two explicit tests above
      continue
    rc, o = _c_decode(data)
    if _parses(Msg, data):
      back = Msg.FromString(data)
      assert rc == 0, data
      _expect(o, back.debugging.check_points_range, back.fftw.planning_rigor, back.max_batch_size,
              back.points_range)
      ok += 1
    else:
      assert rc == 3, data
  assert 400 < ok < 5000, ok


def test_op_desc_from_attrs_restates_the_op_constructors():
  import ctypes
  from tensorflow_nufft import _lib
  lib = _lib.lib()
  err = ctypes.create_string_buffer(256)

  def make(op, tt, fd, tol, prec, data):
    d = _lib.OpDesc()
    rc = lib.nufft_hip_op_desc_from_attrs(ctypes.byref(d), op, tt, fd, tol, prec, data, len(data or b''), err, 256)
    return rc, d, err.value.decode()

  o = tfft.Options(); o.max_batch_size = 4; o.points_range = tfft.PointsRange.INFINITE
  rc, d, _ = make(_lib.OP_NUFFT, b'type_1', b'backward', 1e-5, 4, o.to_proto().SerializeToString())
  assert rc == 0 and (d.op_type, d.transform_type, d.fft_direction, d.precision) == (0, 1, 1, 4)
  assert d.tol == 1e-5 and d.options.max_batch_size == 4 and d.options.points_range == 2
  assert d.source_ndim == 0 and d.points_ndim == 0 and d.grid_shape_len == 0
  rc, d, _ = make(_lib.OP_NUFFT, b'type_2', b'forward', 1e-6, 8, b'')
  assert rc == 0 and (d.transform_type, d.fft_direction, d.options.points_range) == (2, -1, 0)
  # Interp / Spread: fixed types, no options attr (nufft_kernels.cc:590-621): library defaults
  rc, d, _ = make(_lib.OP_INTERP, None, None, 1e-6, 4, None)
  assert rc == 0 and d.transform_type == 2 and d.options.points_range == 1
  rc, d, _ = make(_lib.OP_SPREAD, None, None, 1e-6, 4, None)
  assert rc == 0 and d.transform_type == 1 and d.options.points_range == 1
  # the reference's message for bytes that do not parse (nufft_kernels.cc:584-585)
  rc, d, msg = make(_lib.OP_NUFFT, b'type_1', b'forward', 1e-6, 4, b'\x18')
  assert rc == 3 and msg == 'Unable to parse options string.'
  rc, _, msg = make(_lib.OP_NUFFT, b'type_3', b'forward', 1e-6, 4, b'')
  assert rc == 3 and 'transform_type' in msg
  rc, _, msg = make(_lib.OP_NUFFT, b'type_1', b'sideways', 1e-6, 4, b'')
  assert rc == 3 and 'fft_direction' in msg
  rc, _, msg = make(_lib.OP_NUFFT, b'type_1', b'forward', 1e-6, 2, b'')
  assert rc == 3 and 'precision' in msg
  rc, _, msg = make(9, b'type_1', b'forward', 1e-6, 4, b'')
  assert rc == 3
