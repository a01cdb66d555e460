"""The drop-in boundary as the reference's *_eval.py scripts reach it (no GPU): the scripts' own import statements with
``dropin/`` first on the path, the scripts' own classifier loading (``create_model`` = ``torch.load`` of a whole pickled
module, audio_models/create_model.py:8-17) on a pickle of the REFERENCE's ``M5Net.M5`` class, and the lowering of the
objects the scripts build themselves (torchaudio ``Compose``, un-pickled ConvNets)."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PICKLE = os.path.join(ROOT, "tests", "golden", "ref_m5_module.pt")


def _run(code, with_reference):
    paths = [os.path.join(ROOT, "dropin"), ROOT] + ([REF] if with_reference else [])
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(paths))
    cwd = REF if with_reference else ROOT
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=cwd, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


SCRIPT_IMPORTS = f"""
    import sys, types
    from unittest.mock import MagicMock
    for m in ("torchaudio", "torchaudio.transforms", "torchsde", "librosa", "librosa.display", "statsmodels",
              "statsmodels.stats", "statsmodels.stats.proportion"):
        sys.modules.setdefault(m, MagicMock())        # third-party packages the scripts import at top level; absent here
    # adaptive_attack_eval.py:64,88,99-100 and certified_robustness_eval.py:71-73,87, verbatim
    from audio_models.create_model import *
    from acoustic_system import AcousticSystem
    from diffusion_models.diffwave_sde import *
    from diffusion_models.diffwave_ddpm import create_diffwave_model
    from robustness_eval.certified_robust import *
    import torch
    for obj in (AcousticSystem, RevDiffWave, create_diffwave_model, RobustCertificate, create_model):
        assert obj.__module__.startswith(("audiopure_amd.", "audio_models.create_model")), (obj, obj.__module__)
    import audio_models.create_model as cm
    assert "dropin" in cm.__file__, cm.__file__
    Classifier = create_model({PICKLE!r})
    assert type(Classifier).__module__ == "audiopure_amd.audio_models.M5.M5Net", type(Classifier).__module__
    assert Classifier._get_name() == "M5" and not Classifier.training       # adaptive_attack_eval.py:90
    from audiopure_amd import _native as N
    try:
        Classifier(torch.zeros(1, 1, 16000))
    except N.NativeError as e:                                             # loud, and about the device -- not AttributeError
        assert "HIP device" in str(e) or "CPU" in str(e), e
    else:
        raise SystemExit("CPU forward did not raise")
    AS_MODEL = AcousticSystem(classifier=Classifier, transform=None, defender=None)
    AS_MODEL.eval()
    print("OK")
"""


def test_script_imports_resolve_to_the_native_path_without_the_reference():
    assert "OK" in _run(SCRIPT_IMPORTS, with_reference=False)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout absent")
def test_script_imports_resolve_to_the_native_path_from_inside_the_reference_checkout():
    """cwd = the reference checkout, whose own ./audio_models/M5 create_model.py would put first on sys.path."""
    out = _run(SCRIPT_IMPORTS + """
    import robustness_eval.black_box_attack as bb          # the reference's own module, found through extend_path
    assert bb.NES.__module__ == "audiopure_amd.robustness_eval._NES", bb.NES.__module__
    from diffusion_models.diffwave_ddpm import DiffWave    # transfer_attack_eval.py:9
    assert DiffWave.__module__ == "audiopure_amd.diffusion_models.diffwave_ddpm"
    from audio_models.RCNN_KWS import *                    # kws_adaptive_attack_eval.py:71
    assert KWSModel.__module__ == "audiopure_amd.audio_models.RCNN_KWS.model", KWSModel.__module__
    Classifier = KWSModel(in_size=40)                      # :74
    from diffusion_models.improved_diffusion_sde import *  # adaptive_attack_eval.py:103
    assert RevImprovedDiffusion.__module__ == "audiopure_amd.diffusion_models.improved_diffusion_sde"
    """, with_reference=True)
    assert "OK" in out


def test_reference_pickled_m5_unpickles_into_a_working_native_module():
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        sys.modules.pop("M5Net", None)
        m = torch.load(PICKLE, weights_only=False)
    finally:
        sys.path.remove(os.path.join(ROOT, "dropin"))
    from audiopure_amd.audio_models.M5.M5Net import M5
    assert isinstance(m, M5) and "_native" not in m.__dict__               # restored without __init__
    assert m._native is None and m._key is None                            # class-level defaults
    import io
    buf = io.BytesIO()
    torch.save(m, buf)                                                     # and it pickles again (no ctypes handle inside)


def test_lowering_of_script_built_front_ends_and_classifiers():
    from fake_torchaudio import AmplitudeToDB, Compose, MelSpectrogram, kws_wave2spect, script_wave2spect
    from synth_convnets import vgg19_bn
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.convnet import NativeConvNet
    from audiopure_amd.lowering import lower_classifier, lower_transform
    from audiopure_amd.robustness_eval.certified_robust import RobustCertificate
    from audiopure_amd.transforms import MelSpecDB, MelSpecDBHTK
    for n_mels in (32, 40):
        t = lower_transform(script_wave2spect(n_mels))
        assert type(t) is MelSpecDB and t.n_mels == n_mels
    t = lower_transform(kws_wave2spect(40))
    assert type(t) is MelSpecDBHTK and t.n_mels == 40
    # anything that is not exactly one of the scripts' pipelines is left alone (never approximated)
    odd = Compose([MelSpectrogram(n_fft=1024, hop_length=512, n_mels=32, norm="slaney", pad_mode="constant",
                                  mel_scale="slaney"), AmplitudeToDB("power")])
    assert lower_transform(odd) is odd
    odd2 = Compose([MelSpectrogram(n_fft=2048, hop_length=512, n_mels=32, norm="slaney", pad_mode="constant",
                                   mel_scale="slaney"), AmplitudeToDB("power", top_db=80.0)])
    assert lower_transform(odd2) is odd2
    vgg = vgg19_bn(10, 1, width_div=8).eval()
    sysm = AcousticSystem(classifier=vgg, transform=script_wave2spect(32), defender=None)
    assert isinstance(sysm.classifier, NativeConvNet) and sysm.classifier.module is vgg and type(sysm.transform) is MelSpecDB
    assert sysm.classifier._get_name() == "VGG"
    rc = RobustCertificate(classifier=vgg, transform=script_wave2spect(32), denoiser=None)
    assert isinstance(rc.classifier, NativeConvNet) and type(rc.transform) is MelSpecDB
    lin = torch.nn.Linear(4, 2)
    assert lower_classifier(lin) is lin


def test_reference_class_instances_are_rebuilt_natively():
    """A classifier constructed from the reference's own class (not un-pickled through dropin/) is rebuilt as the native one."""
    import torch.nn as nn
    from audiopure_amd import synth
    from audiopure_amd.lowering import lower_classifier

    class M5(nn.Module):                       # attribute tree of M5Net.py:4-20 (stand-in; torch layers only)
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1, self.pool1 = nn.Conv1d(1, 32, 80, 16), nn.BatchNorm1d(32), nn.MaxPool1d(4)
            self.conv2, self.bn2, self.pool2 = nn.Conv1d(32, 32, 3), nn.BatchNorm1d(32), nn.MaxPool1d(4)
            self.conv3, self.bn3, self.pool3 = nn.Conv1d(32, 64, 3), nn.BatchNorm1d(64), nn.MaxPool1d(4)
            self.conv4, self.bn4, self.pool4 = nn.Conv1d(64, 64, 3), nn.BatchNorm1d(64), nn.MaxPool1d(4)
            self.fc1 = nn.Linear(64, 10)

    src = M5().eval()
    src.load_state_dict({k: torch.from_numpy(v) for k, v in synth.m5_state_dict(10).items()})
    nat = lower_classifier(src)
    assert type(nat).__module__ == "audiopure_amd.audio_models.M5.M5Net" and not nat.training
    for k, v in src.state_dict().items():
        assert torch.equal(nat.state_dict()[k], v)


def test_launcher_and_sitecustomize_beat_a_same_named_module_next_to_the_script(tmp_path):
    """`python script.py` puts the script's directory first on sys.path; a checkout-local acoustic_system.py there must not
    shadow the drop-in, through either entry (dropin/run.py, or PYTHONPATH + sitecustomize)."""
    (tmp_path / "acoustic_system.py").write_text("class AcousticSystem: pass\n")
    (tmp_path / "M5Net.py").write_text("class M5: pass\n")
    script = tmp_path / "fake_eval.py"
    script.write_text("from acoustic_system import AcousticSystem\nimport M5Net\n"
                      "print(AcousticSystem.__module__, M5Net.M5.__module__)\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "dropin", "run.py"), str(script)], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["audiopure_amd.acoustic_system", "audiopure_amd.audio_models.M5.M5Net"], r.stdout
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT])
    r = subprocess.run([sys.executable, str(script)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["audiopure_amd.acoustic_system", "audiopure_amd.audio_models.M5.M5Net"], r.stdout
