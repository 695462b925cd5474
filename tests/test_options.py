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
