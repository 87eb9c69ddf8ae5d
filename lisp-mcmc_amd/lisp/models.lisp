;;;; models.lisp -- model designators: what :function accepts instead of a closure.
;;; The reference's :function is (lambda (x &key m b &allow-other-keys) ...)
;;; (mcmc-fitting.lisp:1134-1137).  A closure cannot run on the GPU, so a function is named
;;; by an enumerated device model (formulas in include/mhx.h) plus the KEYS it reads, in the
;;; model's local order.  Keys are looked up in the walker's parameter plist, so the functions
;;; of a global fit share parameters by naming the same key (README "Global Parameter Fitting").
(in-package #:mcmc-fitting-amd)

(defstruct model
  (id 0 :type fixnum)
  (keys nil :type list)
  (shape nil :type list)
  (expr nil))                           ; C-syntax body for id 7 (MHX_MODEL_EXPR), see expr.lisp

(defun poly-model (&rest keys)
  "f = c0 + c1 x + ... ; (poly-model :b :m) is (+ b (* m x)), mcmc-fitting.lisp:1186"
  (make-model :id 0 :keys keys))

(defun line-model (&optional (b :b) (m :m))
  (poly-model b m))

(defun gauss-peaks-model (bg-keys peak-keys)
  "BG-KEYS: background polynomial keys; PEAK-KEYS: ((:a1 :mu1 :w1) (:a2 :mu2 :w2) ...)"
  (make-model :id 1 :keys (append bg-keys (apply #'append peak-keys))
              :shape (list (length bg-keys) (length peak-keys))))

(defun lorentz-peaks-model (bg-keys peak-keys)
  (make-model :id 2 :keys (append bg-keys (apply #'append peak-keys))
              :shape (list (length bg-keys) (length peak-keys))))

(defun lorder-mixed-bg-model (&key (scale :scale) (linewidth :linewidth) (x0 :x0) (mix :mix)
                                (bg0 :bg0) (bg1 :bg1))
  "the six keys of test.lisp:16-17"
  (make-model :id 3 :keys (list scale linewidth x0 mix bg0 bg1)))

(defun exp-decay-model (&key (a :a) (tau :tau) (c :c))
  (make-model :id 4 :keys (list a tau c)))

(defun sinusoid-model (&key (a :a) (omega :omega) (phi :phi) (c :c))
  (make-model :id 5 :keys (list a omega phi c)))

(defun pvoigt2-model (a b0 b1 mu1 w1 eta1 mu2 w2 eta2 rho c2)
  (make-model :id 6 :keys (list a b0 b1 mu1 w1 eta1 mu2 w2 eta2 rho c2)))

;;; likelihood designators: symbols with the reference's names
(defparameter *likelihood-ids*
  '((log-liklihood-normal . 0) (log-liklihood-normal-weighted . 0)
    (log-liklihood-normal-cutoff . 1) (log-liklihood-poisson . 2)
    (:normal . 0) (:normal-cutoff . 1) (:poisson . 2)))

;;; what create-log-liklihood-function-amd returns (expr.lisp): the closure's body as C text
(defstruct likelihood-spec
  (expr "" :type string))

(defun likelihood-id (designator)
  (let ((d (if (null designator) 'log-liklihood-normal designator)))
    (or (and (likelihood-spec-p d) 3)
        (cdr (assoc d *likelihood-ids*))
        (error 'mhx-error :code -5
                          :message (format nil "unsupported :log-liklihood ~s (a closure cannot ~
                                                cross to the GPU; see INTEGRATION.md)" d)))))

;;; prior designators
(defstruct prior-bounds-spec
  (bounds nil :type list)
  (body-expr nil))                      ; C-syntax prior body, nil = bounds-total (expr.lisp)

(defun prior-bounds (&rest key-low-high)
  "(prior-bounds '(:x -10 10) '(:y 100 200)) stands for a prior whose body is the
bounds-total of (prior-bounds-let ((:x -10 10) (:y 100 200)) bounds-total),
mcmc-fitting.lisp:346-369."
  (make-prior-bounds-spec :bounds key-low-high))

(defun log-prior-flat (&rest ignore)
  (declare (ignore ignore))
  0d0)
