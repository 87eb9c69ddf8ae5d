;;;; package.lisp
;;; The reference package is :mcmc-fitting / nickname :mfit (package.lisp:3-5 of the
;;; reference).  This one is :mcmc-fitting-amd / :mfit-amd so both can be loaded side by
;;; side; code written against the reference switches with
;;;   (rename-package :mcmc-fitting-amd :mcmc-fitting '(:mfit))
;;; or a :local-nicknames entry ((:mfit :mcmc-fitting-amd)).
(defpackage #:mcmc-fitting-amd
  (:nicknames #:mfit-amd)
  (:use #:cl)
  (:export
   ;; walker API (reference: mcmc-fitting.lisp:465, 480, 544, 581, 861, 1176)
   #:walker #:walker-p #:walker-function #:walker-param-keys #:walker-param-style
   #:walker-data #:walker-data-error #:walker-log-liklihood #:walker-log-prior
   #:walker-length #:walker-age #:walker-last-step #:walker-most-likely-step #:walker-walk
   #:walker-step #:make-walker-step #:walker-step-prob #:walker-step-params
   #:walker-create #:mcmc-fit
   #:walker-adaptive-steps #:walker-adaptive-steps-full #:walker-many-steps
   #:walker-take-step #:walker-take-step-injected #:walker-get #:walker-modify #:walker-destroy
   #:walker-save #:walker-load #:diagonal-covariance
   #:mfit-walker-estop #:request-stop
   ;; likelihood / prior designators
   #:log-liklihood-normal #:log-liklihood-normal-weighted #:log-liklihood-normal-cutoff
   #:log-liklihood-poisson #:log-prior-flat #:prior-bounds
   ;; model designators (what replaces a Lisp closure as :function)
   #:model #:make-model #:poly-model #:line-model #:gauss-peaks-model #:lorentz-peaks-model
   #:lorder-mixed-bg-model #:exp-decay-model #:sinusoid-model #:pvoigt2-model
   ;; arbitrary closures / prior bodies, compiled at run time (expr.lisp)
   #:expr-model #:prior-bounds-let-amd #:form->c #:bounds-total
   #:create-log-liklihood-function-amd #:log-normal
   ;; engine-level extras
   #:walker-n-chains #:walker-chain-status #:walker-kernel-name #:mhx-error #:mhx-error-code #:mhx-error-message))
