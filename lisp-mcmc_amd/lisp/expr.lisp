;;;; expr.lisp -- reference-style closures and prior-bounds-let bodies as device expressions.
;;; (expr-model '(lambda (x &key m b &allow-other-keys) (+ b (* m x)))) walks the quoted form
;;; and produces the C-syntax expression mhx_set_function_expr takes (include/mhx.h); libmhx
;;; compiles it for gfx950 with hiprtc into the same fused kernels - or, when the body IS a
;;; polynomial background plus Gaussian / Lorentzian peaks, serves it with that enumerated model's
;;; kernels (recognised below the C ABI: csrc/mhx_expr.cpp; nothing to do here).  The walk mirrors
;;; lisp-mcmc_amd/sexpr.py, which is what the test-suite exercises (no Lisp implementation
;;; exists in the build environment).
(in-package #:mcmc-fitting-amd)

(defun mangle-symbol (sym)
  "Lisp symbol / keyword -> C identifier: downcase, non-alphanumerics -> _"
  (let* ((s (string-downcase (symbol-name sym)))
         (out (map 'string (lambda (c) (if (or (alphanumericp c) (char= c #\_)) c #\_)) s)))
    (if (or (zerop (length out)) (digit-char-p (char out 0)))
        (concatenate 'string "k_" out)
        out)))

(defun c-number (x)
  "a Lisp real -> C literal text; rationals keep their quotient form"
  (cond ((integerp x) (format nil "~d.0" x))
        ((rationalp x) (format nil "(~d.0/~d.0)" (numerator x) (denominator x)))
        (t (let ((*read-default-float-format* 'double-float))
             ;; with the default format bound to double-float, 1d-5 prints as 1.0e-5
             (prin1-to-string (coerce x 'double-float))))))

(defparameter *unary-functions*
  '((exp . "exp") (log . "log") (sqrt . "sqrt") (sin . "sin") (cos . "cos") (tan . "tan")
    (atan . "atan") (tanh . "tanh") (abs . "abs") (floor . "floor")))

(defparameter *comparisons*
  '((< . "<") (> . ">") (<= . "<=") (>= . ">=") (= . "==") (/= . "!=")))

(defun join (strings separator)
  (with-output-to-string (s)
    (loop for (str . rest) on strings
          do (write-string str s)
             (when rest (write-string separator s)))))

(defun form->c (form &optional rename)
  "FORM: the body of a model closure or of a prior-bounds-let; RENAME: alist symbol -> C name"
  (labels ((sym (s)
             (cond ((assoc s rename) (cdr (assoc s rename)))
                   ((eq s 'pi) "3.14159265358979323846")
                   ((eq s 't) "1.0")
                   ((null s) "0.0")
                   (t (mangle-symbol s))))
           (tr (f)
             (cond ((realp f) (c-number f))
                   ((symbolp f) (sym f))
                   ((atom f) (error "form->c: cannot translate ~s" f))
                   (t (let ((op (car f)) (a (cdr f)))
                        (case op
                          ((+ *) (if a
                                     (format nil "(~a)" (join (mapcar #'tr a) (format nil " ~a " op)))
                                     (if (eq op '+) "0.0" "1.0")))
                          (- (if (cdr a)
                                 (format nil "(~a)" (join (mapcar #'tr a) " - "))
                                 (format nil "(-~a)" (tr (first a)))))
                          (/ (if (cdr a)
                                 (format nil "(~a)" (join (mapcar #'tr a) " / "))
                                 (format nil "(1.0 / ~a)" (tr (first a)))))
                          (1+ (format nil "(~a + 1.0)" (tr (first a))))
                          (1- (format nil "(~a - 1.0)" (tr (first a))))
                          ;; an integer exponent is repeated multiplication (intexp order); a
                          ;; float literal with an integral value - the (expt q 2d0) of
                          ;; mcmc-fitting.lisp:377 - is libm's pow there, within an ulp of the
                          ;; product, and the product is what the device forms
                          (expt (let ((p (second a)))
                                  (cond ((and (integerp p) (<= (abs p) 64))
                                         (format nil "ipow(~a, ~d)" (tr (first a)) p))
                                        ((and (floatp p) (= p (fround p)) (<= (abs p) 64))
                                         (format nil "ipow(~a, ~d)" (tr (first a)) (round p)))
                                        ((and (floatp p) (= p 0.5))
                                         (format nil "sqrt(~a)" (tr (first a))))
                                        (t (format nil "pow(~a, ~a)" (tr (first a)) (tr p))))))
                          ((max min)
                           (reduce (lambda (acc v) (format nil "~(~a~)(~a, ~a)" op acc (tr v)))
                                   (cdr a) :initial-value (tr (first a))))
                          (if (format nil "((~a) ? ~a : ~a)" (test (first a)) (tr (second a))
                                      (if (cddr a) (tr (third a)) "0.0")))
                          ;; (elt x 0), (elt x 1), (aref x 1): the components of a vector-valued
                          ;; independent variable (mcmc-fitting.lisp:1136-1137) - columns xcol0,
                          ;; xcol1 of mhx_set_dataset_cols
                          ((elt aref svref)
                           (if (and (symbolp (first a)) (equal (sym (first a)) "x") (integerp (second a)))
                               (format nil "xcol~d" (second a))
                               (error 'mhx-error :code -5
                                                 :message (format nil "~s: only a literal index into the independent variable" f))))
                          (the (tr (second a)))
                          ((coerce float) (tr (first a)))
                          (t (let ((fn (cdr (assoc op *unary-functions*))))
                               (cond ((and fn (= (length a) 1))
                                      (format nil "~a(~a)" fn (tr (first a))))
                                     ((and (eq op 'log) (= (length a) 2))
                                      (format nil "(log(~a) / log(~a))" (tr (first a)) (tr (second a))))
                                     ;; (log-normal x mu sigma), mcmc-fitting.lisp:372-377,
                                     ;; expanded; matched by name, whatever package it was read in
                                     ((and (symbolp op) (string= (symbol-name op) "LOG-NORMAL")
                                           (= (length a) 3))
                                      (destructuring-bind (x mu sigma) a
                                        (tr `(+ (* -1/2 (log (* 2 pi))) (* -1 (log ,sigma))
                                                (* -1/2 (expt (/ (- ,x ,mu) ,sigma) 2d0))))))
                                     (t (error 'mhx-error :code -5
                                                          :message (format nil "operator ~s is not supported in a device expression" op)))))))))))
           (test (f)
             (if (consp f)
                 (let ((op (car f)) (a (cdr f)))
                   (cond ((assoc op *comparisons*)
                          (let ((vals (mapcar #'tr a)) (c (cdr (assoc op *comparisons*))))
                            (if (cdr vals)
                                (format nil "(~a)"
                                        (join (loop for (p q) on vals while q
                                                    collect (format nil "(~a ~a ~a)" p c q))
                                              " && "))
                                "1")))
                         ((eq op 'and) (if a (format nil "(~a)" (join (mapcar #'test a) " && ")) "1"))
                         ((eq op 'or) (if a (format nil "(~a)" (join (mapcar #'test a) " || ")) "0"))
                         ((eq op 'not) (format nil "(!~a)" (test (first a))))
                         (t (format nil "(~a != 0.0)" (tr f)))))
                 (format nil "(~a != 0.0)" (tr f)))))
    (tr form)))

(defun lambda-keys (lambda-list)
  "the &key names of (x &key a (b 0d0) &allow-other-keys), in order"
  (let ((keys nil) (in-keys nil))
    (dolist (item (cdr lambda-list) (nreverse keys))
      (let ((name (if (consp item) (car item) item)))
        (cond ((eq name '&key) (setf in-keys t))
              ((and (symbolp name) (char= (char (symbol-name name) 0) #\&)) (setf in-keys nil))
              (in-keys (push name keys)))))))

(defun expr-model (lambda-form)
  "A model designator from a QUOTED reference-style closure:
(expr-model '(lambda (x &key m b &allow-other-keys) (+ b (* m x))))"
  (destructuring-bind (lambda-sym lambda-list &rest body) lambda-form
    (declare (ignore lambda-sym))
    (let* ((body (remove-if (lambda (b) (and (consp b) (eq (car b) 'declare))) body))
           (keys (lambda-keys lambda-list)))
      (unless (= (length body) 1)
        (error 'mhx-error :code -5 :message "the lambda body must be one expression"))
      (make-model :id 7
                  :keys (mapcar (lambda (k) (intern (symbol-name k) :keyword)) keys)
                  :shape nil
                  :expr (form->c (first body) (list (cons (first lambda-list) "x")))))))

(defun create-log-liklihood-function-amd (lambda-form)
  "The reference's create-log-liklihood-function (mcmc-fitting.lisp:402-416) for a QUOTED
3-argument closure; evaluates to a :log-liklihood DESIGNATOR:
(create-log-liklihood-function-amd
  '(lambda (y model error) (declare (ignore error)) (- (* y (log model)) model)))
The log-likelihood is the sum of its values over the points.  ERROR is the point's sigma, as
the reference's docstring defines it.  The walker's :function must be an expr-model."
  (destructuring-bind (lambda-sym lambda-list &rest body) lambda-form
    (declare (ignore lambda-sym))
    (let ((body (remove-if (lambda (b) (and (consp b) (eq (car b) 'declare))) body)))
      (unless (and (= (length lambda-list) 3) (every #'symbolp lambda-list))
        (error 'mhx-error :code -5
                          :message "the function must accept 3 arguments: y, model, error"))
      (unless (= (length body) 1)
        (error 'mhx-error :code -5 :message "the lambda body must be one expression"))
      (make-likelihood-spec
       :expr (form->c (first body) (list (cons (first lambda-list) "y")
                                         (cons (second lambda-list) "model")
                                         (cons (third lambda-list) "error")))))))

(defmacro prior-bounds-let-amd ((&rest keys-low-high) &body body)
  "Same shape as the reference's prior-bounds-let (mcmc-fitting.lisp:346-369), but evaluates to a
prior DESIGNATOR: (prior-bounds-let-amd ((:x -10 10) (:y 100 200)) (+ bounds-total (if (> x y) -1e9 0e0)))"
  `(make-prior-bounds-spec
    :bounds (list ,@(mapcar (lambda (klh) `(list ,(first klh) ,(second klh) ,(third klh)))
                            keys-low-high))
    :body-expr ,(progn
                  (when (cdr body)
                    (error 'mhx-error :code -5 :message "the prior body must be one expression"))
                  (form->c (first body) (list (cons (intern "BOUNDS-TOTAL") "bounds_total"))))))
