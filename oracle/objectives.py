"""CPU restatement of the HSC and BCE objectives (test infrastructure; see oracle/__init__.py).

Follows reference `src/eoe/training/hsc.py:12-21` (HSCTrainer.compute_anomaly_score / .loss) and
`src/eoe/training/bce.py:15-20` (BCETrainer.compute_anomaly_score / .loss).  torch-CPU fp32; autograd gives
the gradients, and the closed forms below (`*_grad`) are what the HIP backward kernels implement.
"""
import torch


def hsc_dists(features: torch.Tensor) -> torch.Tensor:
    # hsc.py:13,18  dists = sqrt(norm(f, p=2, dim=1)**2 + 1) - 1   (pseudo-Huber distance to the origin)
    nrm = torch.sqrt((features * features).sum(dim=1))
    return torch.sqrt(nrm * nrm + 1) - 1


def hsc_score(features: torch.Tensor) -> torch.Tensor:
    # hsc.py:12-15  scores = 1 - exp(-dists)
    return 1 - torch.exp(-hsc_dists(features))


def hsc_losses(features: torch.Tensor, labels: torch.Tensor, nominal_label: int = 0) -> torch.Tensor:
    # hsc.py:18-20  per-sample loss: dists for nominal samples, -log(scores + 1e-9) otherwise
    d = hsc_dists(features)
    s = 1 - torch.exp(-d)
    return torch.where(labels == nominal_label, d, -torch.log(s + 1e-9))


def hsc_loss(features: torch.Tensor, labels: torch.Tensor, nominal_label: int = 0) -> torch.Tensor:
    # hsc.py:21  losses.mean()
    return hsc_losses(features, labels, nominal_label).mean()


def hsc_loss_grad(features: torch.Tensor, labels: torch.Tensor, nominal_label: int = 0,
                  inv_count: float = None) -> torch.Tensor:
    """closed-form d(mean loss)/d(features): f * coef_i, coef_i = dl/dd * 1/sqrt(|f|^2+1) / N
    (SURVEY.md section 8a row A6)."""
    n = features.shape[0]
    inv = (1.0 / n) if inv_count is None else inv_count
    ss = (features * features).sum(dim=1)
    root = torch.sqrt(ss + 1)
    d = root - 1
    e = torch.exp(-d)
    dl_dd = torch.where(labels == nominal_label, torch.ones_like(d), -e / (1 - e + 1e-9))
    return features * (dl_dd / root * inv)[:, None]


def bce_score(features: torch.Tensor, nominal_label: int = 0) -> torch.Tensor:
    # bce.py:15-17  sigmoid(features).squeeze(); 1 - score if nominal_label != 0
    s = torch.sigmoid(features).squeeze()
    return s if nominal_label == 0 else (1 - s)


def bce_losses(features: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    # numerically stable form of -[y log sig(x) + (1-y) log(1-sig(x))], what
    # torch.nn.functional.binary_cross_entropy_with_logits computes (bce.py:20)
    x = features.reshape(features.shape[0], -1).squeeze(1) if features.dim() > 1 else features
    y = labels.to(x.dtype)
    return torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-torch.abs(x)))


def bce_loss(features: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    # bce.py:19-20  mean reduction
    return bce_losses(features, labels).mean()


def bce_loss_grad(features: torch.Tensor, labels: torch.Tensor, inv_count: float = None) -> torch.Tensor:
    n = features.shape[0]
    inv = (1.0 / n) if inv_count is None else inv_count
    x = features.reshape(n)
    return ((torch.sigmoid(x) - labels.to(x.dtype)) * inv).reshape(features.shape)


# ---------------------------------------------------------------------------------------------------------
# the other objectives of the TRAINER registry
# ---------------------------------------------------------------------------------------------------------
def dsad_losses(features: torch.Tensor, labels: torch.Tensor, nominal_label: int = 0) -> torch.Tensor:
    """`src/eoe/training/dsad.py:17-21`: |f|^2 for nominal samples, 1 / (|f|^2 + 1e-9) for the others"""
    d = (features * features).sum(dim=1)
    return torch.where(labels == nominal_label, d, 1.0 / (d + 1e-9))


def dsad_loss(features, labels, nominal_label: int = 0):
    return dsad_losses(features, labels, nominal_label).mean()


def dsad_loss_grad(features, labels, nominal_label: int = 0, inv_count: float = None):
    """closed form of d(mean loss)/df: 2 f for nominal rows, -2 f / (|f|^2 + 1e-9)^2 for the others"""
    n = features.shape[0]
    inv = (1.0 / n) if inv_count is None else inv_count
    d = (features * features).sum(dim=1, keepdim=True)
    coef = torch.where((labels == nominal_label).unsqueeze(1), torch.full_like(d, 2.0), -2.0 / (d + 1e-9) ** 2)
    return features * coef * inv


def dsvdd_center(batch_features, eps: float = 1e-1) -> torch.Tensor:
    """`src/eoe/training/dsvdd.py:10-22`: mean over the batches of the per-batch mean feature of the nominal samples, entries
    closer to zero than eps pushed to +-eps (zero stays zero)"""
    center = torch.stack([bf.mean(dim=0) for bf in batch_features]).mean(dim=0, keepdim=True).clone()
    neg = (center.abs() < eps) & (center < 0)
    pos = (center.abs() < eps) & (center > 0)
    center[neg] = -eps
    center[pos] = eps
    return center


def dsvdd_score(features, center):
    """`dsvdd.py:24-25` (also the per-sample loss, `:26-27`)"""
    return ((features - center) ** 2).sum(dim=-1)


def dsvdd_loss(features, center):
    return dsvdd_score(features, center).mean()


def dsvdd_loss_grad(features, center, inv_count: float = None):
    inv = (1.0 / features.shape[0]) if inv_count is None else inv_count
    return 2.0 * (features - center) * inv


def focal_losses(logits: torch.Tensor, labels: torch.Tensor, gamma: float = 2.0, eps: float = 1e-7) -> torch.Tensor:
    """`src/eoe/training/focal.py:11-24`: (1 - pt)^gamma * bce with pt = clamp(exp(-bce), eps, 1 - eps)"""
    bce = bce_losses(logits, labels)
    pt = torch.exp(-bce).clamp(eps, 1.0 - eps)
    return (1.0 - pt) ** gamma * bce


def focal_loss(logits, labels, gamma: float = 2.0, eps: float = 1e-7):
    return focal_losses(logits, labels, gamma, eps).mean()


def focal_loss_grad(logits, labels, gamma: float = 2.0, eps: float = 1e-7, inv_count: float = None):
    """closed form: with b = bce, b' = sigmoid(x) - y, pt = exp(-b): dF/dx = (1-pt)^g b' + g (1-pt)^(g-1) pt b' b inside the
    clamp, (1-pt)^g b' where the clamp is active (its derivative is zero there)"""
    x = logits.reshape(-1)
    y = labels.reshape(-1).to(x.dtype)
    n = x.shape[0]
    inv = (1.0 / n) if inv_count is None else inv_count
    b = bce_losses(logits, labels).reshape(-1)
    db = torch.sigmoid(x) - y
    raw = torch.exp(-b)
    pt = raw.clamp(eps, 1.0 - eps)
    inside = (raw >= eps) & (raw <= 1.0 - eps)
    g = (1.0 - pt) ** gamma * db + torch.where(inside, gamma * (1.0 - pt) ** (gamma - 1.0) * pt * db * b, torch.zeros_like(b))
    return (g * inv).reshape(logits.shape)


# ---------------------------------------------------------------------------------------------------------
# CLIP text-prompt objective: training/clip.py:66-103
# ---------------------------------------------------------------------------------------------------------
def clip_logits(image_features: torch.Tensor, text_features: torch.Tensor) -> torch.Tensor:
    # clip.py:84-85  image_features / norm;  100 * f @ T^T
    f = image_features / image_features.norm(dim=-1, keepdim=True)
    return 100.0 * f @ text_features.T


def clip_losses(image_features, labels, text_features, nominal_label: int = 0, leave_one_out: bool = False) -> torch.Tensor:
    # clip.py:85-101: log_softmax, anomalous samples pick the last prompt, nominal ones prompt 0 (one_vs_rest) or the best of
    # the class prompts (leave_one_out); samples with any other label keep loss 0; sign flipped
    sim = clip_logits(image_features, text_features).log_softmax(dim=-1)
    anom = 1 - nominal_label
    loss = torch.zeros_like(sim[:, 0])
    loss = torch.where(labels == anom, sim[:, -1], loss)
    nom = sim[:, :-1].max(-1)[0] if leave_one_out else sim[:, 0]
    loss = torch.where(labels == nominal_label, nom, loss)
    return -loss


def clip_loss(image_features, labels, text_features, nominal_label: int = 0, leave_one_out: bool = False):
    return clip_losses(image_features, labels, text_features, nominal_label, leave_one_out).mean()      # clip.py:102


def clip_score(image_features, center):
    # clip.py:66-79: the text features are normalised again; anomaly score = softmax probability of the last prompt
    t = center / center.norm(dim=-1, keepdim=True)
    return clip_logits(image_features, t).softmax(dim=-1)[:, -1]
