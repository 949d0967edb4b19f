"""Inference API — mirror of the reference's ``model/pred_func.py:18-184`` for the hot path.

Same names and argument meaning (``load_genconvit``, ``preprocess_frame``, ``pred_vid``,
``max_prediction_value``, ``real_or_fake``, ``df_face``, ``face_rec``, ``extract_frames``,
``is_video``, ``set_result``, ``store_result``).  ``prediction.py`` star-imports this module and
relies on ``torch``/``os``/``np`` coming along, so they stay module globals.  The heavy CPU-side
dependencies (dlib, face_recognition, decord) are imported lazily inside the video / face
functions so the model path imports on a box without them; cv2 is not needed any more (the crop +
INTER_AREA resize of ``face_rec`` runs on the device, row N4).
"""
import os

import numpy as np
import torch

from .. import _lib, synth
from .config import load_config
from .genconvit import GenConViT

device = "cuda" if torch.cuda.is_available() else "cpu"


def load_genconvit(config, net, ed_weight, vae_weight, fp16, arch_type="original", use_attention=True,
                   use_residual=True):
    """reference model/pred_func.py:18-64.  ``arch_type='v2'`` (an experiment fork with a different,
    incompatible architecture) is out of scope of this build and rejected explicitly."""
    device_str = "cuda" if torch.cuda.is_available() else "cpu"
    print(f"Using device: {device_str}")
    if arch_type == "v2":
        raise NotImplementedError("GenConViTV2 is an experiment fork outside this build's scope (SURVEY.md §2 #9)")
    model = GenConViT(config, ed=ed_weight, vae=vae_weight, net=net, fp16=fp16)
    model.to(device)
    model.eval()
    if fp16:
        model.half()
    return model


_MEAN = torch.tensor(synth.IMAGENET_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
_STD = torch.tensor(synth.IMAGENET_STD, dtype=torch.float32).view(1, 3, 1, 1)


def preprocess_frame(frame):
    """uint8 (N,224,224,3) -> normalised fp32 NCHW (reference :95-108 with the 'vid' transform of
    dataset/loader.py:63-65,77), vectorised over the batch instead of a per-frame Python loop."""
    u8 = torch.as_tensor(np.asarray(frame))
    if torch.cuda.is_available() and u8.dtype == torch.uint8:
        # row N1: ship the uint8 crops (4x fewer H2D bytes than fp32) and normalise on the device
        return _lib.preprocess(u8.to(device))
    df_tensor = u8.float().permute((0, 3, 1, 2)) / 255.0
    return (df_tensor - _MEAN) / _STD


def pred_vid(df, model):
    """reference :111-120: sigmoid over logits, mean over rows, argmax."""
    with torch.no_grad():
        p = next(model.parameters())
        if df.device != p.device:
            df = df.to(p.device)
        return max_prediction_value(torch.sigmoid(model(df).squeeze()))


def pred_vids(dfs, model, max_batch=128):
    """Row N3: the reference calls the model once per video with <= 15 frames
    (prediction.py:231-266); here the face crops of MANY videos are concatenated into batches of up to
    ``max_batch`` frames, run through one forward each, and voted per video on the device.  Returns
    ``[(y, y_val), ...]`` with the exact ``pred_vid`` semantics per video."""
    results = [None] * len(dfs)
    order = [i for i, d in enumerate(dfs) if len(d) >= 1]
    p = next(model.parameters())
    nets = 2 if getattr(model, "net", "genconvit") not in ("ed", "vae") else 1
    i = 0
    with torch.no_grad():
        while i < len(order):
            group, n = [], 0
            while i < len(order) and (not group or n + len(dfs[order[i]]) <= max_batch):
                group.append(order[i])
                n += len(dfs[order[i]])
                i += 1
            batch = torch.cat([dfs[g].to(p.device) for g in group])
            offs = torch.tensor([0] + list(np.cumsum([len(dfs[g]) for g in group])), dtype=torch.int32)
            logits = model(batch)
            means = _lib.vote_segments(logits, batch.shape[0], nets, offs).cpu()
            for k, g in enumerate(group):
                m = means[k]
                results[g] = (int(torch.argmax(m).item()),
                              m[0].item() if m[0] > m[1] else abs(1 - m[1]).item())
    return results


def max_prediction_value(y_pred):
    """reference :123-131 (the device-side reduction is ``genconvit_amd._lib.vote``)."""
    mean_val = torch.mean(y_pred, dim=0)
    return (
        torch.argmax(mean_val).item(),
        mean_val[0].item() if mean_val[0] > mean_val[1] else abs(1 - mean_val[1]).item(),
    )


def real_or_fake(prediction):
    return {0: "REAL", 1: "FAKE"}[prediction ^ 1]


def extract_frames(video_file, frames_nums=15):
    from decord import VideoReader, cpu
    vr = VideoReader(video_file, ctx=cpu(0))
    step_size = max(1, len(vr) // frames_nums)
    return vr.get_batch(list(range(0, len(vr), step_size))[:frames_nums]).asnumpy()


def face_locations(frames):
    """The detector call of the reference's face_rec (:71-76): dlib's CNN / HOG detector through ``face_recognition``
    on the CPU (third-party; imported lazily).  Returns rows (frame index, top, right, bottom, left), at most
    ``len(frames)`` of them in frame order — the reference stops filling ``temp_face`` there (:78,88-89)."""
    import dlib
    import face_recognition
    mod = "cnn" if dlib.DLIB_USE_CUDA else "hog"
    boxes = []
    for i, frame in enumerate(frames):
        bgr = np.ascontiguousarray(frame[..., ::-1])          # cv2.cvtColor(frame, cv2.COLOR_RGB2BGR) (:72)
        for loc in face_recognition.face_locations(bgr, number_of_times_to_upsample=0, model=mod):
            if len(boxes) < len(frames):
                boxes.append((i, *loc))
    return boxes


def crop_faces(frames, boxes, size=224):
    """Row N4: crop + ``cv2.INTER_AREA`` resize of every box on the MI355X (``gcv_face_crop_resize``); the RGB<->BGR
    swaps the reference wraps around the resize (:72,86) cancel.  uint8 (n,size,size,3) on the device."""
    fr = torch.as_tensor(np.ascontiguousarray(frames)).to(device)
    return _lib.face_crop_resize(fr, boxes, size)


def face_rec(frames, p=None, klass=None, locate=None):
    """reference :67-92.  Detection stays third-party CPU code (``locate``, default ``face_locations`` above); the crops
    are cut and resized on the device and come back as the uint8 array the reference returns."""
    boxes = (locate or face_locations)(frames)
    if len(boxes) == 0:
        return [], 0
    boxes = boxes[:len(frames)]
    return crop_faces(frames, boxes).cpu().numpy(), len(boxes)


def df_face(vid, num_frames, net):
    img = extract_frames(vid, num_frames)
    face, count = face_rec(img)
    return preprocess_frame(face) if count > 0 else []


def is_video(vid):
    return os.path.isfile(vid) and vid.endswith((".avi", ".mp4", ".mpg", ".mpeg", ".mov"))


def set_result():
    return {"video": {"name": [], "pred": [], "klass": [], "pred_label": [], "correct_label": []}}


def store_result(result, filename, y, y_val, klass, correct_label=None, compression=None):
    v = result["video"]
    v["name"].append(filename)
    v["pred"].append(y_val)
    v["klass"].append(klass.lower())
    v["pred_label"].append(real_or_fake(y))
    if correct_label is not None:
        v["correct_label"].append(correct_label)
    if compression is not None:
        v["compression"].append(compression)
    return result
